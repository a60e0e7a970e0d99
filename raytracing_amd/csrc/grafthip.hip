// grafthip.hip -- the launchRays() path of the ray tracer on AMD Instinct MI355X (gfx950, wave64).
//
// One lane traces one pixel; one wave owns one 8x8 pixel tile; waves pull tiles from a shared cursor until
// the band is done.  Scene geometry ({cx,cy,cz,r*r} per sphere, {p,n} per plane) is staged once per
// workgroup into LDS and read by broadcast (all lanes of a wave test the same shape at the same time);
// materials and lights stay in global memory (read once per shaded hit).  Every arithmetic step is IEEE
// double in the reference's order: this file must be compiled with -ffp-contract=off so that no mul/add
// pair is fused (a fused multiply-add changes hit masks in the last bit).
//
// The recursion of shootRay (reference mainTracing.c:300-372) is a per-lane state machine with ONE
// closest-hit site, so lanes that trace primary, shadow and reflection rays share the hot loop.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is loaded when a second GPU is asked for

#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include <algorithm>
#include <array>
#include <utility>

#include "grafthip.h"

#include "trace_types.h"
#include "closest_hit.h"
#include "pack_tables.h"
#include "host_job.h"
#include "host_steps.h"

using namespace graft;

#ifndef GRAFT_SAMPLE_TABLE
#define GRAFT_SAMPLE_TABLE 1
#endif
#ifndef GRAFT_SHARED_RCP
#define GRAFT_SHARED_RCP 1
#endif

// Device memory of the entry point that allocates nothing (graftRenderRowsNoAlloc): every GPU the library runs on
// holds one copy of this array, at an address of its own, until the process ends.  Host code never uses the name
// as an address: graftInit asks the runtime for the array's device address (hipGetSymbolAddress, with that device
// current) and keeps it in Device::scratch.
static __device__ __align__(256) unsigned char scratch[(size_t)4 << 20];

namespace {

// SCREEN: 0 = exact double loop, 1 = single-precision screen per sphere, 2 = two-level ball screen,
// 3 = two-level screens per ray family (cones for primary and shadow rays, balls for the rest; LDS only),
// 4 = large scenes: three-level screens per ray family (an entry per 8 groups on top), the cone families' member
//     sections left in global memory;
// 5 = level 4 with a uniform grid walked lane by lane in place of the ball family (closestHitGrid);
// 6 = candidate lists: per tile for primary rays, per (sphere, light) for shadow rays (closestHitLists), everything in
//     LDS; scenes without mirrors;
// 7 = candidate lists kept in global memory (spheres' screen table and tile lists in LDS), and closestHitGrid for the
//     rays behind a mirror.
// LDSGEOM: the table the hot loop reads (screen table, or the doubles when SCREEN == 0) is staged in LDS.
// EXACTLDS (SCREEN > 0 only): the exact doubles are staged in LDS as well, behind the screen table; otherwise the
// rare exact tests read them from global memory.
// TRANSP: some shape is transparent: every shaded hit may spawn a refraction ray besides the mirror ray, so the
// per-lane stack holds a tree walk (wider entries) instead of a chain.
// MANYCURSORS: tickets come from kCursors counters instead of one (launches of cheap tiles, see below).  A
// separate instantiation because the expensive-tile kernels lose 1-3 % to any change of this code (measured).
// MIRRORS: some shape reflects (or is transparent): a shaded hit may have children, so the lane keeps a stack of
// levels.  Without it (the reference only recurses `if (reflectivity > 0 && shouldReflect == 1)`, mainTracing.c:330,
// or for a transparent shape, :347) a ray's value is its hit's own shading and goes straight into the pixel's sum.
// EXTRAS = false: a launch that asks for neither the index mask nor the ray counts (what launchRays itself produces is
// the frame alone); the mask ray, the per-lane counts and their registers are compiled out.
template <int AVOID, bool LDSGEOM, int SCREEN, bool EXACTLDS, bool TRANSP, bool MANYCURSORS, bool MIRRORS = true, bool EXTRAS = true>
__global__ void __launch_bounds__(kLargeBlock) traceKernel(const TraceParams P) {
    int32_t *const hitIndexOut = EXTRAS ? P.hitIndex : nullptr;
    unsigned long long *const countersOut = EXTRAS ? P.counters : nullptr;
    static_assert(MIRRORS || !TRANSP, "the tree walk needs the stack");
    constexpr bool LISTS = SCREEN == 6 || SCREEN == 7;   // candidate lists
    constexpr bool GLOBALLISTS = SCREEN == 7;             // ... read from global memory
    if (LDSGEOM) {
        if (LISTS) {
            float *dst = (float *)ldsGeom;
            for (int i = threadIdx.x; i < P.ldsHeadFloats; i += blockDim.x) dst[i] = P.geomF[i];
        } else if (SCREEN == 4 || SCREEN == 5) {
            float *dst = (float *)ldsGeom;
            for (int i = threadIdx.x; i < P.ldsHeadFloats; i += blockDim.x) dst[i] = P.geomF[i];
            float *tail = dst + 4 * (size_t)P.ballMembersLds;
            for (int i = threadIdx.x; i < P.ldsTailFloats; i += blockDim.x) tail[i] = P.geomF[P.ldsTailSrc + i];
        } else if (SCREEN) {
            float *dst = (float *)ldsGeom;
            for (int i = threadIdx.x; i < P.filterFloats; i += blockDim.x) dst[i] = P.geomF[i];
            if (EXACTLDS) {
                double *dd = (double *)((char *)ldsGeom + P.exactLdsOffset);
                for (int i = threadIdx.x; i < P.geomDoubles; i += blockDim.x) dd[i] = P.geom[i];
            }
        } else {
            for (int i = threadIdx.x; i < P.geomDoubles + kGeomPad; i += blockDim.x) ldsGeom[i] = P.geom[i];
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int lx = lane & (kTile - 1), ly = lane >> 3;
    // Kernels that draw single tiles from one counter and have candidate lists are only launched for supersampled frames
    // whose table of sample scale factors is in LDS (launchTrace): both are known when they are compiled.
#ifdef GRAFT_TABLE_AT_RUN_TIME  // (A/B builds: asked per launch, as in the other kernels)
    constexpr bool kTabled = false;
#else
    constexpr bool kTabled = SCREEN >= 6 && GRAFT_SAMPLE_TABLE && !MANYCURSORS;
#endif
    const bool aliasing = kTabled || P.aliasing;
    const bool tabled = kTabled || (SCREEN >= 6 && GRAFT_SAMPLE_TABLE && !MANYCURSORS && P.sampleTableOff >= 0);
    const int nSamples = aliasing ? P.numPasses * P.numPasses : 1;
    // Ray statistics (GraftCounters): the reference's closest-hit queries and ray-shape evaluations for these pixels,
    // and the queries answered here (shadow rays whose answer cannot matter are not shot, see the lights' loop).  Per
    // lane and tile three small counts -- queries other than shadow rays (the reference makes exactly those too), shaded
    // hits (the reference shoots numLights shadow rays for each), queries answered; they are summed over the wave at
    // the end of a tile, and only in launches that asked for counters.
    unsigned long long rayCount = 0, testCount = 0, tracedCount = 0;  // wave totals (uniform)
    unsigned nOther = 0, nHits = 0, nAnswered = 0;
#ifdef GRAFT_STAMPS  // diagnostic build only: shader-clock cycles per wave in the closest-hit call and in total
    unsigned long long stampHit = 0, stampRounds = 0, stampLoop = 0;
    const unsigned long long stampBegin = clock64();
#endif

    // Candidate lists: where the wave's tile list starts (in LDS, one list per wave of the workgroup; in chunks of four
    // entries, counted from the pool's start at level 6 and from the start of LDS at level 7) and how long the current
    // tile's is (kListAll: every sphere)
    const int tileListPos = !LISTS ? 0 : ((GLOBALLISTS ? P.listTileOff : P.listTileOff - P.listPoolOff) >> 3) + (int)(threadIdx.x >> 6) * (kTileListCap / 4);
    int tileListLen = 0;

    // rayIntersectAvoid for one ray per lane (mainTracing.c:128-153): picks the walk the job was built for; returns
    // how many shapes the ray's avoid list removes from the brute-force count.  `from`: the shape a shadow ray starts on.
    auto closest = [&](const double o[3], const double d[3], int mode, int li, int depth, int from, double tLimit, Hit &h) -> unsigned {
        int avoidShape = -1;
        const unsigned char *avoidRow = nullptr;
        unsigned avoided = 0;
        if (AVOID != 0 && mode == M_SHADOW) {
            const DevLight &L = P.lights[li];
            avoided = (unsigned)L.avoidNum;
            if (AVOID == 1) avoidShape = L.avoidFirst;
            if (AVOID == 2 && L.avoidNum > 0) avoidRow = P.avoidMask + (size_t)li * P.numShapes;
        }
        if (SCREEN == 7 && mode != M_SHADOW && depth > 0) {
            // a ray that left a mirror passes through no point known ahead of time: the uniform grid
            const double s1 = fabs(o[0]) + fabs(o[1]) + fabs(o[2]);
            if (s1 < P.familyScale) {
                const float negThreshold = -(float)((64.0 * 5.9604644775390625e-08 * s1 * s1 + 1e-30) * 1.000001);
                closestHitGrid<AVOID>(P, o[0], o[1], o[2], d[0], d[1], d[2], -1, nullptr, negThreshold, INFINITY, h);
            } else {
                closestHit<AVOID, false>(P, o[0], o[1], o[2], d[0], d[1], d[2], -1, nullptr, h);
            }
        } else if (LISTS && AVOID == 0) {
            // Level 6 in a scene whose lights avoid nothing: one walk for every ray.  A ray that starts beyond the
            // scale the lists were built for (none does in a job that qualifies) looks at every sphere, behind the
            // single-precision screen alone, whose threshold never proves anything for coordinates it cannot bound.
            const double s1 = fabs(o[0]) + fabs(o[1]) + fabs(o[2]);
            const float negThreshold = s1 < 1e15 ? -(float)((64.0 * 5.9604644775390625e-08 * s1 * s1 + 1e-30) * 1.000001) : -INFINITY;
            int pos = tileListPos, len = tileListLen, self = -1;
            bool tile = true;
            if (mode == M_SHADOW) {
                const unsigned *index = (const unsigned *)((GLOBALLISTS ? (const char *)P.geomF : (const char *)ldsGeom) + P.listIndexOff);
                const unsigned w = index[from * P.numLights + li];
                pos = (int)(w & 0xfffffu);
                len = (int)(w >> 20);
                self = from;
                tile = false;
            }
            if (len == (int)kListAll || !(s1 < P.familyScale)) { pos = 0; len = P.numSpheres; self = -1; tile = false; }  // (the pool starts with every sphere)
            closestHitLists<GLOBALLISTS>(P, o[0], o[1], o[2], d[0], d[1], d[2], pos, len, tile, self, negThreshold, h);
        } else if (SCREEN) {
            const double s1 = fabs(o[0]) + fabs(o[1]) + fabs(o[2]);
            // (level 6: every direction comes out of vecUnit -- unit length or the zero vector -- and every origin is a
            // screen point or a point of a sphere, all finite in a job that qualifies; the test on the origin stays)
            const bool tame = LISTS || (s1 < 1e15 && fabs(d[0]) <= 1.0000001 && fabs(d[1]) <= 1.0000001 && fabs(d[2]) <= 1.0000001);
            if (tame && (SCREEN == 1 || s1 < (SCREEN >= 3 ? P.familyScale : 1e5))) {
                const float negThreshold = -(float)((64.0 * 5.9604644775390625e-08 * s1 * s1 + 1e-30) * 1.000001);
                if (LISTS) {
                    // the lane's list: the occluders of (sphere the ray starts on, light), or what the tile can see
                    // (a shadow ray's own sphere is not on its list: closestHitLists looks at it first)
                    int pos = tileListPos, len = tileListLen, self = -1;
                    bool tile = true;
                    if (mode == M_SHADOW) {
                        const unsigned *index = (const unsigned *)((GLOBALLISTS ? (const char *)P.geomF : (const char *)ldsGeom) + P.listIndexOff);
                        const unsigned w = index[from * P.numLights + li];
                        pos = (int)(w & 0xfffffu);
                        len = (int)(w >> 20);
                        self = from;
                        tile = false;
                        if (AVOID == 1 && from == avoidShape) self = -1;
                        if (AVOID == 2 && avoidRow && avoidRow[from]) self = -1;
                    }
                    if (len == (int)kListAll) { pos = 0; len = P.numSpheres; self = -1; tile = false; }  // (the pool starts with every sphere)
                    closestHitLists<GLOBALLISTS>(P, o[0], o[1], o[2], d[0], d[1], d[2], pos, len, tile, self, negThreshold, h);
                } else if (SCREEN >= 3) {
                    // which fixed point the ray's line passes through decides the table it is screened with
                    int family = P.ballFamily;
                    if (mode == M_SHADOW) family = P.lightFamilyBase + li;
                    else if ((mode == M_INDEX || depth == 0) && P.cameraFamily >= 0) family = P.cameraFamily;
                    // level 5: only the coherent rays of a pixel's first level walk the families' tables
                    if (SCREEN == 5 && (depth > 0 || family < 0))
                        closestHitGrid<AVOID>(P, o[0], o[1], o[2], d[0], d[1], d[2], avoidShape, avoidRow, negThreshold, tLimit, h);
                    else
                        closestHitFamilies<AVOID, EXACTLDS, SCREEN >= 4, SCREEN != 5>(P, o[0], o[1], o[2], d[0], d[1], d[2], avoidShape, avoidRow, family, negThreshold, h);
                } else if (SCREEN == 2) closestHitGrouped<AVOID, LDSGEOM, EXACTLDS>(P, o[0], o[1], o[2], d[0], d[1], d[2], avoidShape, avoidRow, negThreshold, h);
                else closestHitScreened<AVOID, LDSGEOM, EXACTLDS>(P, o[0], o[1], o[2], d[0], d[1], d[2], avoidShape, avoidRow, negThreshold, h);
            } else {  // exact loop over the doubles in global memory
                closestHit<AVOID, false>(P, o[0], o[1], o[2], d[0], d[1], d[2], avoidShape, avoidRow, h);
            }
        } else {
            closestHit<AVOID, LDSGEOM, true>(P, o[0], o[1], o[2], d[0], d[1], d[2], avoidShape, avoidRow, h);
        }
        return avoided;
    };
    // one finished pixel: number p of the band (row-major)
    auto storeColor = [&](size_t p, const double acc[3]) {
        if (P.format == GRAFT_OUT_F64_RGB) {
            double *f = (double *)P.frame + 3 * p;
            f[0] = acc[0]; f[1] = acc[1]; f[2] = acc[2];
        } else if (P.format == GRAFT_OUT_F32_RGBA) {
            float4 px = make_float4((float)acc[0], (float)acc[1], (float)acc[2], 1.0f);
            ((float4 *)P.frame)[p] = px;
        } else {
            unsigned w = quant8(acc[0]) | (quant8(acc[1]) << 8) | (quant8(acc[2]) << 16) | 0xff000000u;
            ((unsigned *)P.frame)[p] = w;
        }
    };

    // Tiles are handed out in tickets of `tilesPerTicket` consecutive tiles drawn from shared counters.  One address
    // takes about 10^8 atomics per second, which cheap tiles -- one sample per pixel -- exceed even at four tiles
    // per ticket; counters in different cache lines add up.
    // One counter (expensive tiles): a wave's first ticket is its own number (the host starts the counter at the
    // number of waves); the next ticket is drawn when a ticket's first tile starts and only read after its last
    // tile, so the atomic's round trip overlaps the work.  With candidate lists many tiles are cheap after all (an
    // empty tile list: background only) and a frame of them runs at the counter's pace: tickets hold four tiles
    // until the end of the launch is near, and one tile from there on.
    // kCursors counters (cheap tiles): counter c hands out tickets c, c + kCursors, ...; a wave starts at counter
    // (wave number mod kCursors) and moves on to the next counter when one is used up, until all are.  The next
    // ticket is drawn ahead only when there are plenty per wave (P.prefetchTickets): a wave holding a spare ticket
    // would otherwise starve the others.
    int nextTicket = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));  // one counter: lane 0's draw
    int cursor = MANYCURSORS ? nextTicket % kCursors : 0;
    int usedUp = 0;  // counters found empty so far (wave-uniform)
    if (MANYCURSORS && P.prefetchTickets && lane == 0) nextTicket = atomicAdd(P.tileCursor + cursor * kCursorStride, 1);
    int tile = 0, tileEnd = 0;
    int listEnd = 0;  // MANYCURSORS with candidate lists: the tile list in LDS covers tiles [.., listEnd) of this ticket
    for (;;) {
        if (tile >= tileEnd) {
            if (MANYCURSORS) {
                const int numTickets = (P.numTiles + P.tilesPerTicket - 1) / P.tilesPerTicket;
                if (!P.prefetchTickets && lane == 0) nextTicket = atomicAdd(P.tileCursor + cursor * kCursorStride, 1);
                int ticket = __builtin_amdgcn_readfirstlane(nextTicket) * kCursors + cursor;
                while (ticket >= numTickets && ++usedUp < kCursors) {  // this counter is used up: try the next one
                    cursor = cursor + 1 == kCursors ? 0 : cursor + 1;
                    if (lane == 0) nextTicket = atomicAdd(P.tileCursor + cursor * kCursorStride, 1);
                    ticket = __builtin_amdgcn_readfirstlane(nextTicket) * kCursors + cursor;
                }
                if (ticket >= numTickets) break;  // every wave gets here: the counters only grow
                tile = ticket * P.tilesPerTicket;
                tileEnd = min(tile + P.tilesPerTicket, P.numTiles);
                listEnd = tile;
                if (P.prefetchTickets && lane == 0) nextTicket = atomicAdd(P.tileCursor + cursor * kCursorStride, 1);
            } else {
                // (the first P.coarseTickets tickets hold tilesPerTicket tiles each, the tickets after them -- the end of
                // the launch, where waves must not be left holding several expensive tiles -- one tile)
                const int ticket = __builtin_amdgcn_readfirstlane(nextTicket);
                const int fine = ticket - P.coarseTickets;
                tile = fine < 0 ? ticket * P.tilesPerTicket : P.coarseTickets * P.tilesPerTicket + fine;
                if (tile >= P.numTiles) break;  // every wave reaches this: the counter only grows
                tileEnd = min(tile + (fine < 0 ? P.tilesPerTicket : 1), P.numTiles);
                if (lane == 0) nextTicket = atomicAdd(P.tileCursor, 1);
            }
        }
        const int ty = tile / P.tilesX, tx = tile - ty * P.tilesX;
        const int ix = tx * kTile + lx;
        const int by = ty * kTile + ly;  // row inside the band
        const int iy = P.row0 + by;
        const bool inside = ix < P.nCols && by < P.bandRows;

        const ConstDoubles cam = cameraView(P);
        // What this tile's primary rays can see, worked out once by all lanes together.  Cheap tiles (a ticket of several
        // tiles, few samples per pixel) share one list per run of a ticket's tiles in one row of tiles: the list costs as
        // much as tracing a tile's 64 rays, and a rectangle a few tiles wide sees hardly more spheres than one tile.
        // (Not in the kernels with mirrors: measured 5 % slower there -- their register allocation is tight, and the
        // tile list is a small part of what they do.)
#ifdef GRAFT_LIST_PER_TILE  // (A/B builds: one list per tile whatever the ticket)
        constexpr bool kSharedLists = false;
#else
        constexpr bool kSharedLists = MANYCURSORS && !MIRRORS;
#endif
        if (LISTS && (!kSharedLists || tile >= listEnd)) {
            int lastTx = tx;
            if (kSharedLists) {
                listEnd = min(tileEnd, (ty + 1) * P.tilesX);
                lastTx = listEnd - 1 - ty * P.tilesX;
            }
            const int ix0 = tx * kTile, ix1 = min(lastTx * kTile + kTile, P.nCols) - 1;
            const int iy0 = P.row0 + ty * kTile, iy1 = P.row0 + min(ty * kTile + kTile, P.bandRows) - 1;
            // samples lie at the pixel's offsets plus 0 (the mask ray) .. numPasses sub-steps (mainTracing.c:405-406)
            const double spanX = aliasing ? cam[CAM_SUBX] * (double)P.numPasses : 0.0;
            const double spanY = aliasing ? cam[CAM_SUBY] * (double)P.numPasses : 0.0;
            const int found = buildTileList<GLOBALLISTS>(P, cam, P.colOff[ix0] + fmin(spanX, 0.0), P.colOff[ix1] + fmax(spanX, 0.0),
                                            P.rowOff[iy0] + fmin(spanY, 0.0), P.rowOff[iy1] + fmax(spanY, 0.0),
                                            (unsigned short *)((char *)ldsGeom + P.listTileOff) + (threadIdx.x >> 6) * kTileListCap);
            tileListLen = found < 0 ? (int)kListAll : found;
            // Supersampling: the scale factor of a sample's first getNewPoint step (mainTracing.c:405, :107-108: a
            // division and a square root) depends on (column, k) only, that of the second (:406) on (row, l): all
            // 16 numPasses of them are worked out once per tile, one per lane, and read back per sample.
            if (tabled) {
                double *tab = (double *)((char *)ldsGeom + P.sampleTableOff) + (threadIdx.x >> 6) * 16 * P.numPasses;
                if (lane < kTile * P.numPasses) {
                    const int c = (int)__umulhi((unsigned)lane, P.numPassesRecip), k = lane - c * P.numPasses;
                    const double offX = P.colOff[min(ix0 + c, ix1)] + cam[CAM_SUBX] * (double)(k + 1);
                    const double offY = P.rowOff[min(iy0 + c, iy1)] + cam[CAM_SUBY] * (double)(k + 1);
                    const double scaleX = sqrt(offX * offX / cam[CAM_LRDEN]), scaleY = sqrt(offY * offY / cam[CAM_UDDEN]);
                    tab[lane] = offX < 0 ? -scaleX : scaleX;
                    tab[kTile * P.numPasses + lane] = offY < 0 ? -scaleY : scaleY;
                }
            }
        }
        double acc[3] = {cam[CAM_BACKGROUND], cam[CAM_BACKGROUND + 1], cam[CAM_BACKGROUND + 2]};  // mainTracing.c:395
        int firstIdx = -1;
        if (inside) {
            const double colI = P.colOff[ix], rowJ = P.rowOff[iy];
            const size_t pixel = (size_t)by * P.nCols + ix;
            // per-lane machine state
            int mode, depth = 0, sample = 0, li = 0, sp = 0, hitIdx = -1, litLight = -1;
            double o[3], d[3];                 // current ray
            double loc[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, din[3] = {0, 0, 0};  // shaded hit: point, normal, incoming dir
            double litW = 0;
            // One entry per shaded hit whose children are still being traced: sum so far [3], tint [3], wRefl,
            // and with TRANSP: wRefr, origin [3] and direction [3] of the refraction ray.  Flags: see F_*.
            constexpr int kEntry = TRANSP ? 14 : 7;
            constexpr int kLevels = MIRRORS ? kMaxDepth : 1;  // (without mirrors nothing is ever pushed)
            double stk[kLevels * kEntry];
            unsigned char flags[kLevels];
            int rayReflects = 1;               // shouldReflect of the ray being traced (mainTracing.c:300)
            double tLimit = INFINITY;          // shadow rays: ray parameter of the light (see closestHitGrid)

            // the screen point of sample s (k = s / numPasses outer, l inner: mainTracing.c:401-408)
            auto samplePoint = [&](int s, double out[3]) {
                double xo = colI, yo = rowJ;
                if (aliasing) {
                    // k = s / numPasses by a multiplication (exact below 2^20 samples, host_job.h)
                    const int k = (int)__umulhi((unsigned)s, P.numPassesRecip), l = s - k * P.numPasses;
                    if (tabled) {  // the two steps' scale factors from the tile's table
                        const double *tab = (const double *)((const char *)ldsGeom + P.sampleTableOff) + (threadIdx.x >> 6) * 16 * P.numPasses;
                        const double scaleX = tab[lx * P.numPasses + k], scaleY = tab[(kTile + ly) * P.numPasses + l];
                        for (int c = 0; c < 3; ++c) {
                            const double col = cam[CAM_SCREEN + c] + scaleX * cam[CAM_LR + c];
                            out[c] = col + scaleY * cam[CAM_UD + c];
                        }
                        return;
                    }
                    xo = colI + cam[CAM_SUBX] * (double)(k + 1);
                    yo = rowJ + cam[CAM_SUBY] * (double)(l + 1);
                }
                screenPoint(cam, xo, yo, out);
            };
            auto startSample = [&](int s) {
                samplePoint(s, o);
                d[0] = d[1] = d[2] = 0;
                if (P.projectionType == GRAFT_ORTHOGRAPHIC) unit3(-cam[CAM_VIEW], -cam[CAM_VIEW + 1], -cam[CAM_VIEW + 2], d[0], d[1], d[2]);
                else unit3(o[0] - cam[CAM_POS], o[1] - cam[CAM_POS + 1], o[2] - cam[CAM_POS + 2], d[0], d[1], d[2]);
                depth = 0;
                sp = 0;
                rayReflects = 1;  // launchRays passes shouldReflect = 1, mainTracing.c:391
                mode = M_RADIANCE;
            };
            // Which is the FIRST light that may lie in front of the surface at the hit being shaded?  "Behind" only when
            // the cosine the reference computes is below -1e-10 for certain, far from its rounding.  The lights are gone
            // through from light 0 until one is in front -- neighbouring pixels mostly stop at the same one --, their
            // positions come through scalar loads.  (Jobs whose colours and weights are not all finite -- there
            // 0 * colour is not 0 --, small scenes and jobs with more than 31 lights take every light to be in front.)
            int firstFront = 0;
            auto findFront = [&]() {
                firstFront = 0;
                if (!P.skipLightsBehind || P.numLights > 31) return;
                const ConstDoubles LD = (ConstDoubles)(unsigned long long)P.lights;
                const double nn = nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2];
                int i = 0;
                for (; i < P.numLights; ++i) {
                    const double bx = LD[8 * i + 3] - loc[0], by2 = LD[8 * i + 4] - loc[1], bz = LD[8 * i + 5] - loc[2];
                    const double q = bx * nrm[0] + by2 * nrm[1] + bz * nrm[2];
                    const double bb = bx * bx + by2 * by2 + bz * bz;
                    if (!(q < 0 && q * q > 1e-20 * (bb * nn))) break;
                }
                firstFront = i;  // (numLights: every light lies behind the surface)
            };
            // may one of the lights 0 .. last shine on the hit?
            auto mayShine = [&](int last) { return last >= firstFront; };
            // The stages of a round do not start the lane's next ray themselves: they say which kind it is, and ONE
            // piece of code at the end of the round normalises its direction and sets its origin for all lanes
            // together.  (Lanes of a wave are in different stages: each stage doing this for itself would run the
            // square root and the three divisions of vecUnit up to three times per round.)
            enum { NEXT_NONE = 0, NEXT_SHADOW = 1, NEXT_MIRROR = 2, NEXT_SAMPLE = 3 };

            // maxDepth <= 0: shootRay returns black before it looks at the scene (mainTracing.c:302-306), for every
            // sample; with anti-aliasing the background-seeded sum (:395) takes n additions of zero
            const bool traceNothing = P.maxDepth <= 0;
            if (LISTS && tileListLen == 0 && !traceNothing) {
                // No sphere can be met by any primary ray of this tile (mask ray included): every sample is the
                // background colour (mainTracing.c:314-317), summed as the reference sums it (:419).  The queries
                // count as answered.
                const unsigned queries = (unsigned)nSamples + (hitIndexOut && aliasing ? 1u : 0u);
                nOther += queries;
                nAnswered += queries;
                if (aliasing)
                    for (int s = 0; s < nSamples; ++s) {
                        acc[0] = acc[0] + cam[CAM_BACKGROUND]; acc[1] = acc[1] + cam[CAM_BACKGROUND + 1]; acc[2] = acc[2] + cam[CAM_BACKGROUND + 2];
                    }
                mode = M_DONE;
            } else if (hitIndexOut && (aliasing || traceNothing)) {  // mask ray = the anti-aliasing-off ray of this pixel
                primaryRay(P, colI, rowJ, o, d);
                mode = M_INDEX;
            } else if (traceNothing) {
                mode = M_DONE;
            } else {
                startSample(0);
            }
            if (traceNothing) {
                if (aliasing) { acc[0] = acc[0] + 0.0; acc[1] = acc[1] + 0.0; acc[2] = acc[2] + 0.0; }
                else acc[0] = acc[1] = acc[2] = 0;
            }

#ifdef GRAFT_STAMPS
            const unsigned long long stampLoop0 = clock64();
            const bool loopLeader = lane == (int)__builtin_ctzll(__ballot(1));
#endif
            while (mode != M_DONE) {
#ifdef GRAFT_STAMPS
                const unsigned long long stamp0 = clock64();
#endif
                Hit h;
                const unsigned avoided = closest(o, d, mode, li, depth, hitIdx, mode == M_SHADOW ? tLimit : INFINITY, h);
                nAnswered++;
                if (mode != M_SHADOW) nOther++;  // (a hit's shadow rays are booked when it is found)
                (void)avoided;
#ifdef GRAFT_STAMPS
                const unsigned long long stamp1 = clock64();
                if (lane == (int)__builtin_ctzll(__ballot(1))) {  // one lane of those in this trip books the time
                    stampHit += stamp1 - stamp0;
                    stampRounds++;
                }
#endif

#ifdef GRAFT_STAMPS  // shader-clock shares of the stages behind the query: booked by the first lane present at each mark
                unsigned long long stampMark = clock64();
                auto stampStage = [&](int which) {
                    const unsigned long long now = clock64();
                    if (countersOut && lane == (int)__builtin_ctzll(__ballot(1))) atomicAdd(countersOut + 17 + which, now - stampMark);
                    stampMark = now;
                };
#define GRAFT_STAGE(which) stampStage(which)
#else
#define GRAFT_STAGE(which)
#endif
                bool finalize = false, unwind = false;
                int next = NEXT_NONE;
                double v[3] = {0, 0, 0};
                if (mode == M_INDEX) {
                    firstIdx = h.idx;
                    if (traceNothing) mode = M_DONE;
                    else next = NEXT_SAMPLE;
                } else if (mode == M_RADIANCE) {
                    if (depth == 0 && sample == 0 && !aliasing) firstIdx = h.idx;
                    if (h.idx < 0) {  // mainTracing.c:314-317
                        v[0] = cam[CAM_BACKGROUND]; v[1] = cam[CAM_BACKGROUND + 1]; v[2] = cam[CAM_BACKGROUND + 2];
                        unwind = true;
                    } else {
                        hitIdx = h.idx;
                        loc[0] = h.x; loc[1] = h.y; loc[2] = h.z;
                        din[0] = d[0]; din[1] = d[1]; din[2] = d[2];
                        nrm[0] = nrm[1] = nrm[2] = 0;
                        const double *g = (LDSGEOM && SCREEN == 0) ? (const double *)ldsGeom
                                          : (EXACTLDS || SCREEN == 6) ? (const double *)((const char *)ldsGeom + P.exactLdsOffset) : P.geom;
                        if (h.kind == GRAFT_SPHERE) {  // shapes.c:69-71
                            const double *c = g + 4 * (size_t)h.slot;
                            if (LISTS && GRAFT_SHARED_RCP) unit3Shared(loc[0] - c[0], loc[1] - c[1], loc[2] - c[2], nrm[0], nrm[1], nrm[2]);
                            else unit3(loc[0] - c[0], loc[1] - c[1], loc[2] - c[2], nrm[0], nrm[1], nrm[2]);
                        } else {  // shapes.c:91
                            const double *pn = g + 4 * (size_t)P.numSpheres + 6 * (size_t)h.slot + 3;
                            nrm[0] = pn[0]; nrm[1] = pn[1]; nrm[2] = pn[2];
                        }
                        litW = 0;
                        litLight = -1;
                        // lighting() (mainTracing.c:196-229) shoots a shadow ray per light, first to last, and every
                        // unshadowed light OVERWRITES rgb (:223-225): its result is the last unshadowed light's.  So
                        // the lights are gone through last to first and the first one found unshadowed ends the loop:
                        // what the lights before it would have answered is never looked at by the reference either.
                        nHits++;
                        // A light behind the surface (cosine with the normal below zero) adds nothing whether it is
                        // shadowed or not (diffInt < 0 ? 0, :219); its shadow ray only decides whether the lights
                        // BEFORE it are overwritten.  Once every light still to be asked lies behind the surface
                        // no ray can change the result: they are not asked (mayShine below).
                        li = P.numLights - 1;
                        findFront();
                        if (mayShine(li)) next = NEXT_SHADOW;
                        else finalize = true;
                    }
                } else {  // M_SHADOW: mainTracing.c:210-226
                    const DevLight &L = P.lights[li];
                    bool lit = h.idx < 0;
                    if (!lit) {
                        double bx = L.loc[0] - h.x, by2 = L.loc[1] - h.y, bz = L.loc[2] - h.z;
                        // The reference normalises this vector and asks whether its cosine with the ray is below 0.1.
                        // With q = b.d and |b|^2 the answer is known without the square root and the three divisions
                        // unless the cosine is within half a percent of 0.1 or of 0 (the roundings of either form are
                        // below 1e-15 of the terms): a blocker well before the light has cosine 1, one beyond it -1.
                        const double q = bx * d[0] + by2 * d[1] + bz * d[2];
                        const double bb = bx * bx + by2 * by2 + bz * bz;
                        const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                        const double qq = q * q;
                        if (q > 0 && qq > 0.0101 * (bb * dd) && dd > 0.99 && dd < 1.01 && bb > 1e-280) {
                            lit = false;
                        } else if (q < 0 && qq > 1e-20 * (bb * dd) && bb < 1e300) {
                            lit = true;
                        } else {
                            unit3(bx, by2, bz, bx, by2, bz);
                            lit = dot3(bx, by2, bz, d[0], d[1], d[2]) < 0.1;
                        }
                    }
                    if (lit) {  // direction to the light is recomputed by the reference with identical operations
                        const double w = dot3(d[0], d[1], d[2], nrm[0], nrm[1], nrm[2]);
                        litW = w < 0 ? 0 : w;
                        litLight = li;
                    }
                    if (!lit && mayShine(--li)) next = NEXT_SHADOW;
                    else finalize = true;
                }

                GRAFT_STAGE(0);  // what the query found: hit (normal, lights in front) or shadow result
                if (!MIRRORS && !TRANSP && finalize) {  // mainTracing.c:352-369 for a hit without children
                    const DevShade &m = P.shade[hitIdx];
                    double light[3] = {0, 0, 0};
                    if (litLight >= 0) {  // the last unshadowed light overwrites, mainTracing.c:223-225
                        const DevLight &L = P.lights[litLight];
                        for (int c = 0; c < 3; ++c) {
                            double t = L.color[c] * m.color[c];
                            light[c] = litW * t;
                        }
                    }
                    for (int c = 0; c < 3; ++c) {
                        const double lit = m.wDiff * light[c];
                        const double part = lit + m.ambient[c];
                        double out = m.mirrorZero + part;  // reflectionRGB and refractionRGB stay {0,0,0}
                        out = (0.0 * 0.0) + out;
                        v[c] = out;
                    }
                    unwind = true;
                }
                if ((MIRRORS || TRANSP) && finalize) {  // mainTracing.c:330-369 for the level whose hit is (hitIdx, loc, nrm, din)
                    const DevMaterial &m = P.materials[hitIdx];
                    const double rgb[3] = {m.color[0], m.color[1], m.color[2]};
                    double light[3] = {0, 0, 0};
                    if (litLight >= 0) {  // the last unshadowed light overwrites, mainTracing.c:223-225
                        const DevLight &L = P.lights[litLight];
                        for (int c = 0; c < 3; ++c) {
                            double t = L.color[c] * rgb[c];
                            light[c] = litW * t;
                        }
                    }
                    const bool mirror = MIRRORS && m.reflectivity > 0 && rayReflects == 1;   // :330
                    const bool glass = TRANSP && m.transparency > 0;               // :347, :355
                    const double wDiff = (1 - m.ambientLight) * (1 - m.reflectivity) * (1 - m.transparency);
                    double wRefl = (1 - m.ambientLight) * m.reflectivity, wRefr = 0;
                    if (glass) {
                        wRefl = fresnelWeight(m.refrIndex, nrm, din);
                        wRefr = (1 - wRefl);
                    }
                    double part[3];
                    for (int c = 0; c < 3; ++c) {
                        const double amb = m.ambientLight * rgb[c];
                        const double lit = wDiff * light[c];
                        part[c] = lit + (amb + 0.0);
                    }
                    if (mirror && !glass && depth + 1 >= P.maxDepth) {
                        // The mirror child would return black at once (mainTracing.c:305-306) and there is no other
                        // child: this level's value with the operations of :338, :363-369 on a black reflection,
                        // without a trip through the stack.
                        for (int c = 0; c < 3; ++c) {
                            const double t = (1 - .5) * (1.0 - rgb[c]);  // mainTracing.c:334-336
                            double mir = (rgb[c] + t) * 0.0;
                            mir = wRefl * mir;
                            const double out = mir + part[c];
                            v[c] = (0.0 * 0.0) + out;
                        }
                        unwind = true;
                    } else if (mirror || glass) {  // children to trace: push this level
                        double *lv = stk + kEntry * sp;
                        unsigned char fl = 0;
                        for (int c = 0; c < 3; ++c) {
                            const double t = (1 - .5) * (1.0 - rgb[c]);  // mainTracing.c:334-336
                            lv[3 + c] = rgb[c] + t;
                        }
                        lv[6] = wRefl;
                        if (glass) {
                            if (TRANSP) {
                                lv[7] = wRefr;
                                if (refractionRay(m.refrIndex, loc, din, nrm, lv + 8, lv + 11)) fl |= F_REFRACTION_REFLECTS;
                            }
                            fl |= F_HAS_REFRACTION;
                        }
                        if (mirror) {
                            for (int c = 0; c < 3; ++c) lv[c] = part[c];
                            flags[sp++] = fl | F_MIRROR_PENDING;
                            if (depth + 1 >= P.maxDepth) {  // the child returns -2 at once, mainTracing.c:305-306
                                unwind = true;              // v stays 0
                            } else {  // reflection(), mainTracing.c:180-190
                                const double k2 = 2 * dot3(din[0], din[1], din[2], nrm[0], nrm[1], nrm[2]);
                                d[0] = din[0] - k2 * nrm[0];  // (normalised in place below, as the reference does)
                                d[1] = din[1] - k2 * nrm[1];
                                d[2] = din[2] - k2 * nrm[2];
                                next = NEXT_MIRROR;
                                depth++;
                                rayReflects = 1;  // reflection() passes shouldReflect = 1, :181
                                mode = M_RADIANCE;
                            }
                        } else {  // glass without a mirror ray: reflectionRGB stays {0,0,0}; go straight to refraction
                            for (int c = 0; c < 3; ++c) lv[c] = (wRefl * 0.0) + part[c];
                            flags[sp++] = fl;
                            if (depth + 1 >= P.maxDepth) {
                                unwind = true;  // v stays 0: the refraction child returns -2
                            } else if (TRANSP) {
                                o[0] = lv[8]; o[1] = lv[9]; o[2] = lv[10];
                                d[0] = lv[11]; d[1] = lv[12]; d[2] = lv[13];
                                rayReflects = (fl & F_REFRACTION_REFLECTS) ? 1 : 0;
                                depth++;
                                mode = M_RADIANCE;
                            }
                        }
                    } else {  // no children: reflectionRGB and refractionRGB stay {0,0,0}
                        for (int c = 0; c < 3; ++c) {
                            const double mir = wRefl * 0.0;
                            double out = mir + part[c];
                            out = (wRefr * 0.0) + out;
                            v[c] = out;
                        }
                        unwind = true;
                    }
                }

                GRAFT_STAGE(1);  // shading of a finished hit
                if (unwind) {  // a ray's value v is known: hand it up the tree, innermost first (:338, :363-369)
                    bool launched = false;
                    while (MIRRORS && sp > 0 && !launched) {
                        double *lv = stk + kEntry * (sp - 1);
                        const unsigned char fl = flags[sp - 1];
                        if (fl & F_MIRROR_PENDING) {  // v is the mirror child's value
                            for (int c = 0; c < 3; ++c) {
                                double mir = lv[3 + c] * v[c];
                                mir = lv[6] * mir;
                                lv[c] = mir + lv[c];
                            }
                            if (TRANSP && (fl & F_HAS_REFRACTION)) {  // now the refraction child, same depth
                                flags[sp - 1] = fl & ~F_MIRROR_PENDING;
                                if (sp >= P.maxDepth) {
                                    v[0] = v[1] = v[2] = 0;  // it returns -2 at once
                                } else {
                                    o[0] = lv[8]; o[1] = lv[9]; o[2] = lv[10];
                                    d[0] = lv[11]; d[1] = lv[12]; d[2] = lv[13];
                                    rayReflects = (fl & F_REFRACTION_REFLECTS) ? 1 : 0;
                                    depth = sp;
                                    mode = M_RADIANCE;
                                    launched = true;
                                }
                            } else {
                                for (int c = 0; c < 3; ++c) v[c] = (0.0 * 0.0) + lv[c];
                                sp--;
                            }
                        } else {  // v is the refraction child's value
                            for (int c = 0; c < 3; ++c) {
                                const double rf = (TRANSP ? lv[7] : 0.0) * v[c];
                                v[c] = rf + lv[c];
                            }
                            sp--;
                        }
                    }
                    if (!launched) {
                        if (aliasing) {  // mainTracing.c:419
                            acc[0] = acc[0] + v[0]; acc[1] = acc[1] + v[1]; acc[2] = acc[2] + v[2];
                        } else {
                            acc[0] = v[0]; acc[1] = v[1]; acc[2] = v[2];
                        }
                        sample++;
                        if (sample < nSamples) next = NEXT_SAMPLE;
                        else mode = M_DONE;
                    }
                }

                GRAFT_STAGE(2);  // values handed up, sample finished
                if (next != NEXT_NONE) {  // the lane's next ray
                    double vx, vy, vz;
                    if (next == NEXT_SHADOW) {  // mainTracing.c:200-205
                        const DevLight &L = P.lights[li];
                        vx = L.loc[0] - loc[0]; vy = L.loc[1] - loc[1]; vz = L.loc[2] - loc[2];
                        d[0] = d[1] = d[2] = 0;
                        mode = M_SHADOW;
                    } else if (next == NEXT_MIRROR) {  // reflection(), mainTracing.c:180-190
                        vx = d[0]; vy = d[1]; vz = d[2];
                    } else {  // the sample's primary ray, mainTracing.c:401-417 / :425-437
                        samplePoint(sample, o);
                        if (P.projectionType == GRAFT_ORTHOGRAPHIC) {
                            vx = -cam[CAM_VIEW]; vy = -cam[CAM_VIEW + 1]; vz = -cam[CAM_VIEW + 2];
                        } else {
                            vx = o[0] - cam[CAM_POS]; vy = o[1] - cam[CAM_POS + 1]; vz = o[2] - cam[CAM_POS + 2];
                        }
                        d[0] = d[1] = d[2] = 0;
                        depth = 0;
                        sp = 0;
                        rayReflects = 1;  // launchRays passes shouldReflect = 1, mainTracing.c:391
                        mode = M_RADIANCE;
                    }
                    if (LISTS && GRAFT_SHARED_RCP) unit3Shared(vx, vy, vz, d[0], d[1], d[2]);
                    else unit3(vx, vy, vz, d[0], d[1], d[2]);
                    if (next != NEXT_SAMPLE) {  // vecNudge, mainTracing.c:163-167
                        o[0] = loc[0] + .000001 * d[0];
                        o[1] = loc[1] + .000001 * d[1];
                        o[2] = loc[2] + .000001 * d[2];
                    }
                    if (SCREEN == 5 && next == NEXT_SHADOW) {
                        const DevLight &L = P.lights[li];
                        tLimit = (L.loc[0] - o[0]) * d[0] + (L.loc[1] - o[1]) * d[1] + (L.loc[2] - o[2]) * d[2];
                    }
                }
                GRAFT_STAGE(3);  // next ray set up
            }
#ifdef GRAFT_STAMPS
            if (loopLeader) stampLoop += clock64() - stampLoop0;
#endif
            if (aliasing) {  // mainTracing.c:422-423: counter == numPasses^2
                const double n = (double)nSamples;
                acc[0] = acc[0] / n; acc[1] = acc[1] / n; acc[2] = acc[2] / n;
            }
            if (hitIndexOut) hitIndexOut[pixel] = firstIdx;
            storeColor(pixel, acc);
        }
        if (countersOut) {  // (all lanes: the shuffles read lanes outside the frame too, whose counts are zero)
            unsigned long long other = nOther, hits = nHits, answered = nAnswered;
            for (int off = 32; off > 0; off >>= 1) {
                other += __shfl_down(other, off);
                hits += __shfl_down(hits, off);
                answered += __shfl_down(answered, off);
            }
            other = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(other >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)other);
            hits = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(hits >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)hits);
            answered = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(answered >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)answered);
            rayCount += other + hits * (unsigned)P.numLights;
            testCount += other * (unsigned)P.numShapes + hits * ((unsigned long long)P.numLights * (unsigned)P.numShapes - (unsigned)P.avoidTotal);
            tracedCount += answered;
        }
        nOther = nHits = nAnswered = 0;
        ++tile;
    }
    if (countersOut) {
        for (int off = 32; off > 0; off >>= 1) {
#ifdef GRAFT_STAMPS
            stampHit += __shfl_down(stampHit, off);
            stampRounds += __shfl_down(stampRounds, off);
            stampLoop += __shfl_down(stampLoop, off);
#endif
        }
        if (lane == 0) {
            atomicAdd(countersOut, rayCount);
            atomicAdd(countersOut + 1, testCount);
            atomicAdd(countersOut + 2, tracedCount);
#ifdef GRAFT_STAMPS
            atomicAdd(countersOut + 4, (unsigned long long)(clock64() - stampBegin));
            atomicAdd(countersOut + 5, stampHit);
            atomicAdd(countersOut + 6, stampRounds);
            atomicAdd(countersOut + 7, 1ull);
            atomicAdd(countersOut + 8, stampLoop);
#endif
        }
    }
}

__global__ void probeSqrtKernel(const double *a, double *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sqrt(a[i]);
}
__global__ void probeDivKernel(const double *a, const double *b, double *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] / b[i];
}
__global__ void probeUnitKernel(const double *v, double *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        double x = 0, y = 0, z = 0;  // (left at zero when the length is not positive, vector.c:76)
        unit3Shared(v[3 * i], v[3 * i + 1], v[3 * i + 2], x, y, z);
        out[3 * i] = x; out[3 * i + 1] = y; out[3 * i + 2] = z;
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------

char gError[512] = "no error";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(gError, sizeof gError, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(GRAFT_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// A device buffer that is reused from call to call (an interactive host renders frame after frame with the same
// sizes): grow-only, released by graftShutdown.  hipMalloc + hipFree cost about as much as tracing a 512x512 frame
// of the reference's start-up scene.
struct Reusable {
    void *ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&ptr, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

constexpr int kMaxDevices = 16;
constexpr int kHostCopyRounds = 16;   // rounds of a multi-device frame whose rows leave for the host on their own (= Device::pieceTraced)
constexpr int kMultiReservedCUs = 4;  // CUs a multi-device frame's traces leave to the transfer kernels
constexpr size_t kScratchBytes = (size_t)4 << 20;  // device block of graftRenderRowsNoAlloc: tables + ticket counters

// What the library holds per GPU.
struct Device {
    bool ready = false;
    int id = -1, numCU = 0;
    int physical = -1;  // the HIP device behind it (= id, except for the emulated devices of graftInitDevices)
    hipStream_t stream = nullptr;
    Reusable frame, index, counters;      // device side of graftRenderRows' host buffers
    hipStream_t copyStream = nullptr;     // graftRenderRows: finished pieces of a large frame leave on it while the next are traced
    hipEvent_t pieceTraced[16] = {nullptr};
    hipEvent_t roundMoved[16] = {nullptr};  // graftRenderFrameMulti, root: the transfers of a round have arrived
    Reusable arena;                       // device block of the job graftRenderRows keeps between calls
    std::vector<unsigned char> cacheKey;  // the input that job was packed from
    GraftJob *cacheJob = nullptr;
    std::vector<const void *> configured;  // kernels whose dynamic-LDS limit has been raised on this device
    struct Occupancy { const void *fn; int block; size_t ldsBytes; int perCU; };
    std::vector<Occupancy> occupancy;      // resident workgroups per CU, as the runtime answered
    // graftRenderRowsNoAlloc: this device's copy of `scratch`, a page-locked staging block of the same size and the
    // event that says the last upload from it has finished (all three set up by graftInit, never per call)
    unsigned char *scratch = nullptr;
    unsigned char *scratchStage = nullptr;
    hipEvent_t scratchCopied = nullptr;
    // graftRenderFrameMulti: a second stream (the root receives on it while it traces on `stream`), the event that
    // orders it behind `stream`, this device's rank in the communicator and its pieces' buffers
    hipStream_t moveStream = nullptr;
    hipEvent_t traced = nullptr;
    hipEvent_t frameStart = nullptr, kernelsDone = nullptr, movesDone = nullptr;  // timed: a frame's first kernel queued, its last kernel and its last transfer done
    ncclComm_t comm = nullptr;
    Reusable pieces, pieceIndex;
};

// RCCL, loaded on demand (a process that drives one GPU never needs it)
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) commInitAll = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclGroupStart) groupStart = nullptr;
    decltype(&ncclGroupEnd) groupEnd = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGetErrorString) errorString = nullptr;
};
Rccl gRccl;
int gMultiDevices = 0;  // devices 0 .. gMultiDevices-1 bound by graftInitDevices
// Emulated devices (GRAFT_DIAG_EMULATED_DEVICES, for tests on a machine with fewer GPUs than the code path needs): with
// n > 0 graftInitDevices may bind up to n devices, device d standing for HIP device d mod (GPUs present), each with its
// own streams, buffers and copy of the job; pieces then move to the root with device-to-device copies on the sender's
// transfer stream instead of RCCL send/receive (two ranks of one communicator cannot share a GPU).  Everything else
// of a multi-device frame -- the plan, the buffers and offsets, the ordering of kernels, transfers and copies to the
// host -- is the code that runs on real devices.
int gEmulatedDevices = 0;
bool gMultiEmulated = false;  // the devices bound by graftInitDevices are emulated ones
// what the last multi-device frame measured per device (graftMultiStats)
float gMultiTraceMs[kMaxDevices], gMultiTailMs[kMaxDevices];
int gMultiStatsDevices = 0;

Device gDevices[kMaxDevices];
Device *gBound = nullptr;  // the device of graftInit: the single-device entry points work on it
Settings gSettings;
bool gSettingsRead = false;
TableCache gTables;

// The diagnostic settings start from the environment, read ONCE (at the first graftInit); afterwards only
// graftDiagnosticSet changes them.
void readSettingsOnce() {
    if (gSettingsRead) return;
    gSettingsRead = true;
    if (const char *e = getenv("GRAFT_SCREEN"))
        if (e[0] >= '0' && e[0] <= '0' + kMaxScreen && !e[1]) gSettings.screenCap = e[0] - '0';
    if (const char *e = getenv("GRAFT_GROUP")) gSettings.groupSize = atoi(e);
    if (const char *e = getenv("GRAFT_PER_TICKET")) gSettings.tilesPerTicket = atoi(e) > 0 ? atoi(e) : 0;
    if (const char *e = getenv("GRAFT_RESERVED_CUS")) gSettings.reservedCUs = atoi(e) > 0 ? atoi(e) : 0;
}

typedef void (*TraceFn)(const TraceParams);

// (GRAFT_DEV_KERNELS: a development build that only holds the kernels of the measured workloads -- seconds to
// compile instead of minutes; every other job is refused.)
template <int A, bool M>
TraceFn pickKernelA(bool lds, int screen, bool exactLds, bool mirrors, bool extras) {
#if !defined(GRAFT_STAMPS) && !defined(GRAFT_NO_PLAIN_KERNELS)
    // (the kernels without mask and counts: candidate lists, lights that avoid nothing, no mirrors -- the kernels with
    // mirrors sit at their register cap, and theirs came out 4 % slower than the ones they would replace)
    if (A == 0 && lds && !extras && !mirrors && screen == 6) return traceKernel<0, true, 6, false, false, M, false, false>;
    if (A == 0 && lds && !extras && !mirrors && screen == 7) return traceKernel<0, true, 7, false, false, M, false, false>;
#endif
#ifdef GRAFT_DEV_KERNELS
    if (A != 0 || !lds) return nullptr;
    if (screen == 7) return mirrors ? traceKernel<0, true, 7, false, false, M, true> : traceKernel<0, true, 7, false, false, M, false>;
    if (screen == 6) return traceKernel<0, true, 6, false, false, M, false>;
    if (screen == 5) return traceKernel<0, true, 5, false, false, M>;
    if (screen == 3 && exactLds) return mirrors ? traceKernel<0, true, 3, true, false, M> : traceKernel<0, true, 3, true, false, M, false>;
    if (screen == 0) return traceKernel<0, true, 0, false, false, M>;
    return nullptr;
#else
    if (!lds) return screen == 2 ? traceKernel<A, false, 2, false, false, M> : screen == 1 ? traceKernel<A, false, 1, false, false, M> : traceKernel<A, false, 0, false, false, M>;
    if (screen == 0) return traceKernel<A, true, 0, false, false, M>;
    if (screen == 1) return exactLds ? traceKernel<A, true, 1, true, false, M> : traceKernel<A, true, 1, false, false, M>;
    if (screen == 2) return exactLds ? traceKernel<A, true, 2, true, false, M> : traceKernel<A, true, 2, false, false, M>;
    if (screen == 3 && mirrors) return exactLds ? traceKernel<A, true, 3, true, false, M> : traceKernel<A, true, 3, false, false, M>;
    if (screen == 3) return exactLds ? traceKernel<A, true, 3, true, false, M, false> : traceKernel<A, true, 3, false, false, M, false>;
    if (screen == 4) return traceKernel<A, true, 4, false, false, M>;
    if (screen == 5) return traceKernel<A, true, 5, false, false, M>;
    if (screen == 6) return traceKernel<A, true, 6, false, false, M, false>;  // (packJob gives level 6 to scenes without mirrors only)
    return mirrors ? traceKernel<A, true, 7, false, false, M, true> : traceKernel<A, true, 7, false, false, M, false>;
#endif
}

// Scenes with a transparent shape run the tree-walk kernels, built only for the general avoid mode, screen
// levels 0 and 1, and exact doubles in global memory (the jobs are normalised to that in packJob).
template <bool M>
TraceFn pickKernelM(int avoidMode, bool lds, int screen, bool exactLds, bool transp, bool mirrors, bool extras) {
#ifndef GRAFT_DEV_KERNELS
    if (transp) {
        if (lds) return screen ? traceKernel<2, true, 1, false, true, M> : traceKernel<2, true, 0, false, true, M>;
        return screen ? traceKernel<2, false, 1, false, true, M> : traceKernel<2, false, 0, false, true, M>;
    }
#endif
    return avoidMode == 0 ? pickKernelA<0, M>(lds, screen, exactLds, mirrors, extras)
         : avoidMode == 1 ? pickKernelA<1, M>(lds, screen, exactLds, mirrors, extras) : pickKernelA<2, M>(lds, screen, exactLds, mirrors, extras);
}

TraceFn pickKernel(int avoidMode, bool lds, int screen, bool exactLds, bool transp, bool mirrors, bool manyCursors, bool extras) {
    return manyCursors ? pickKernelM<true>(avoidMode, lds, screen, exactLds, transp, mirrors, extras)
                       : pickKernelM<false>(avoidMode, lds, screen, exactLds, transp, mirrors, extras);
}

}  // namespace

struct GraftJob {
    TraceParams P;
    int avoidMode;
    int screen;   // 0 = exact only, 1 = screen per sphere, 2 = two-level ball screen, 3/4 = per-family two-level screens
    bool lds;
    bool exactLds;  // exact doubles staged in LDS behind the screen table
    bool transp;    // some shape is transparent: tree-walk kernel
    bool mirrors;   // some shape reflects
    size_t ldsBytes;
    int block;
    void *arena;      // the device block holding every table
    bool ownsArena;   // false: the block belongs to a Device (graftRenderRows' cached job)
    Device *dev;
};

namespace {

void describe(GraftJob *job, const PackedJob &J, void *base, bool owns, Device *dev) {
    job->P = bindImage(J, base);
    job->avoidMode = J.avoidMode;
    job->screen = J.screen;
    job->lds = J.lds;
    job->exactLds = J.exactLds;
    job->transp = J.transp;
    job->mirrors = J.mirrors;
    job->ldsBytes = J.ldsBytes;
    job->block = J.block;
    job->arena = base;
    job->ownsArena = owns;
    job->dev = dev;
}

void releaseDevice(Device &D);

// `physical` >= 0: entry `device` of the library's table stands for that HIP device (emulated devices, below).
int initDevice(int device, Device **out, int physical = -1) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (physical < 0) physical = device;
    if (device < 0 || physical >= n || device >= kMaxDevices) return fail(GRAFT_ERR_ARG, "device %d of %d", device, n);
    Device &D = gDevices[device];
    if (!D.ready) {
        readSettingsOnce();
        D.id = device;
        D.physical = physical;
        auto create = [&]() -> int {
            HIP_TRY(hipSetDevice(physical));
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, physical));
            D.numCU = prop.multiProcessorCount;
            HIP_TRY(hipStreamCreateWithFlags(&D.stream, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&D.copyStream, hipStreamNonBlocking));
            for (hipEvent_t &e : D.pieceTraced) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (hipEvent_t &e : D.roundMoved) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            static_assert(sizeof(scratch) == kScratchBytes, "scratch size");
            HIP_TRY(hipGetSymbolAddress((void **)&D.scratch, HIP_SYMBOL(scratch)));  // this device's copy of the array
            HIP_TRY(hipHostMalloc((void **)&D.scratchStage, kScratchBytes, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&D.scratchCopied, hipEventDisableTiming));
            return GRAFT_OK;
        };
        const int rc = create();
        if (rc != GRAFT_OK) {  // nothing half-made stays behind: the next call starts from scratch
            D.ready = true;
            releaseDevice(D);
            return rc;
        }
        D.ready = true;
    }
    *out = &D;
    return GRAFT_OK;
}

void releaseDevice(Device &D) {
    if (!D.ready) return;
    (void)hipSetDevice(D.physical);
    if (D.stream) {
        (void)hipStreamSynchronize(D.stream);
        (void)hipStreamDestroy(D.stream);
    }
    if (D.copyStream) {
        (void)hipStreamSynchronize(D.copyStream);
        (void)hipStreamDestroy(D.copyStream);
    }
    D.copyStream = nullptr;
    for (hipEvent_t &e : D.pieceTraced) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    for (hipEvent_t &e : D.roundMoved) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    if (D.comm && gRccl.commDestroy) (void)gRccl.commDestroy(D.comm);
    D.comm = nullptr;
    if (D.moveStream) (void)hipStreamDestroy(D.moveStream);
    if (D.traced) (void)hipEventDestroy(D.traced);
    if (D.frameStart) (void)hipEventDestroy(D.frameStart);
    if (D.kernelsDone) (void)hipEventDestroy(D.kernelsDone);
    if (D.movesDone) (void)hipEventDestroy(D.movesDone);
    D.moveStream = nullptr;
    D.traced = nullptr;
    D.frameStart = D.kernelsDone = D.movesDone = nullptr;
    D.pieces.release();
    D.pieceIndex.release();
    if (D.scratchCopied) (void)hipEventDestroy(D.scratchCopied);
    if (D.scratchStage) (void)hipHostFree(D.scratchStage);
    D.scratchCopied = nullptr;
    D.scratchStage = nullptr;
    D.scratch = nullptr;
    delete D.cacheJob;
    D.cacheJob = nullptr;
    D.cacheKey.clear();
    D.frame.release();
    D.index.release();
    D.counters.release();
    D.arena.release();
    D.configured.clear();
    D.occupancy.clear();
    D.stream = nullptr;
    D.ready = false;
}

// Queue one trace of rows [row0, row0 + nRows) of `job` on stream `s` of the job's device (which must be current).
int launchTrace(GraftJob *job, int row0, int nRows, int format, void *frame, int32_t *hitIndex, GraftCounters *counters,
                hipStream_t s) {
    Device *D = job->dev;
    TraceParams P = job->P;
    P.row0 = row0; P.bandRows = nRows; P.format = format;
    P.frame = frame; P.hitIndex = hitIndex; P.counters = (unsigned long long *)counters;
    P.tilesX = (int)(((long long)P.nCols + kTile - 1) / kTile);
    const long long tiles = (long long)P.tilesX * (((long long)nRows + kTile - 1) / kTile);
    if (tiles > 0x7fff0000LL) return fail(GRAFT_ERR_ARG, "graftJobTrace: %lld tiles in one launch", tiles);
    P.numTiles = (int)tiles;
    const int wavesPerBlock = job->block / 64;
    // cheap tiles (few samples per pixel) are drawn several at a time and from several counters
    const int samples = P.aliasing ? P.numPasses * P.numPasses : 1;
    // (... and so are supersampled frames with candidate lists whose table of sample scale factors found no room in LDS
    // or would be too long: the single-counter list kernels are compiled for frames that have one)
    const bool manyCursors = samples < 4 || (job->screen >= 6 && P.sampleTableOff < 0);
    TraceFn fn = pickKernel(job->avoidMode, job->lds, job->screen, job->exactLds, job->transp, job->mirrors, manyCursors,
                           hitIndex != nullptr || counters != nullptr);
    if (!fn) return fail(GRAFT_ERR_UNSUPPORTED, "this development build holds no kernel for the job");
    if (job->lds && std::find(D->configured.begin(), D->configured.end(), (const void *)fn) == D->configured.end()) {
        // more dynamic LDS than the default limit: raised once per kernel and device, when the kernel is first used
        HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
        D->configured.push_back((const void *)fn);
    }
    // persistent grid: as many workgroups as the kernel's registers and LDS let a CU hold (asked once per device, kernel,
    // workgroup size and LDS footprint)
    int blocksPerCU = 0;
    for (const Device::Occupancy &o : D->occupancy)
        if (o.fn == (const void *)fn && o.block == job->block && o.ldsBytes == job->ldsBytes) blocksPerCU = o.perCU;
    if (!blocksPerCU) {
        int perCU = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (const void *)fn, job->block, job->ldsBytes));
        blocksPerCU = perCU < 1 ? 1 : perCU;
        if (D->occupancy.size() >= 64) D->occupancy.clear();
        D->occupancy.push_back({(const void *)fn, job->block, job->ldsBytes, blocksPerCU});
    }
    long long wantBlocks = (tiles + wavesPerBlock - 1) / wavesPerBlock;
    // (reserved CUs: a trace that shares its GPU with transfer kernels -- RCCL's send and receive are kernels -- leaves
    // them room to start at once instead of when one of its own persistent workgroups retires)
    const int reserved = gSettings.reservedCUs > 0 ? gSettings.reservedCUs : 0;
    long long maxBlocks = (long long)(D->numCU - reserved > 1 ? D->numCU - reserved : 1) * blocksPerCU;
    const int grid = (int)(wantBlocks < maxBlocks ? wantBlocks : maxBlocks);
    const long long waves = (long long)grid * wavesPerBlock;
    // tiles per ticket: 1 for expensive tiles, 2 or 4 for cheap ones, but never so many that a wave gets fewer than 8 draws
    // (load balance at the end of the launch)
    // (with candidate lists a ticket's tiles share one tile list per row of tiles: longer tickets)
    long long perTicket = samples < 4 ? (job->screen >= 6 && !job->mirrors ? 4 : 2) : 1;
    if (perTicket * waves * 8 > tiles) perTicket = tiles / (waves * 8) < 1 ? 1 : tiles / (waves * 8);
#ifndef GRAFT_COARSE_TILES
#define GRAFT_COARSE_TILES 4
#endif
    // one counter: with candidate lists, tickets of four tiles except for the last 16 tiles per wave
    long long coarse = 0;
    if (!manyCursors && job->screen >= 6 && tiles > waves * 16) {
        perTicket = GRAFT_COARSE_TILES;
        coarse = (tiles - waves * 16) / perTicket;
    }
    if (gSettings.tilesPerTicket > 0) {  // diagnostics: forced ticket size (one counter: for every ticket but the last)
        perTicket = gSettings.tilesPerTicket;
        coarse = manyCursors ? 0 : tiles / perTicket;
    }
    P.tilesPerTicket = (int)perTicket;
    P.coarseTickets = (int)coarse;
    P.prefetchTickets = (tiles + perTicket - 1) / perTicket >= 8 * waves ? 1 : 0;
    if (manyCursors) HIP_TRY(hipMemsetAsync(P.tileCursor, 0, sizeof(int) * kCursors * kCursorStride, s));
    else HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)P.tileCursor, grid * wavesPerBlock, 1, s));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(job->block), job->ldsBytes, s, P);
    HIP_TRY(hipGetLastError());
    return GRAFT_OK;
}

int checkTraceArgs(const char *who, const GraftJob *job, int row0, int nRows, int format, const void *frame) {
    if (!job || !frame) return fail(GRAFT_ERR_ARG, "%s: null job or frame", who);
    if (graftPixelBytes(format) == 0) return fail(GRAFT_ERR_ARG, "%s: unknown format %d", who, format);
    if (row0 < 0 || nRows <= 0 || row0 + nRows > job->P.nRows)
        return fail(GRAFT_ERR_ARG, "%s: rows [%d,+%d) outside frame of %d rows", who, row0, nRows, job->P.nRows);
    return GRAFT_OK;
}

}  // namespace

extern "C" {

GRAFT_API int graftDeviceCount(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        fail(GRAFT_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
        return -GRAFT_ERR_HIP;
    }
    return n;
}

GRAFT_API int graftInit(int device) {
    if (gBound && gBound->id == device) return GRAFT_OK;
    if (gBound) return fail(GRAFT_ERR_STATE, "graftInit: already bound to device %d", gBound->id);
    Device *D = nullptr;
    const int rc = initDevice(device, &D);
    if (rc != GRAFT_OK) return rc;
    gBound = D;
    return GRAFT_OK;
}

GRAFT_API void graftShutdown(void) {
    for (Device &D : gDevices) releaseDevice(D);
    gBound = nullptr;
    gMultiDevices = 0;
    gMultiStatsDevices = 0;
}

GRAFT_API const char *graftLastError(void) { return gError; }

GRAFT_API int graftDiagnosticSet(int which, int value) {
    readSettingsOnce();
    switch (which) {
        case GRAFT_DIAG_SCREEN_CAP:
            if (value < 0 || value > kMaxScreen) return fail(GRAFT_ERR_ARG, "graftDiagnosticSet: screen cap %d", value);
            gSettings.screenCap = value;
            return GRAFT_OK;
        case GRAFT_DIAG_GROUP_SIZE:
            if (value != 0 && (value < 4 || value > 64 || value % 4)) return fail(GRAFT_ERR_ARG, "graftDiagnosticSet: group size %d", value);
            gSettings.groupSize = value;
            return GRAFT_OK;
        case GRAFT_DIAG_TILES_PER_TICKET:
            if (value < 0) return fail(GRAFT_ERR_ARG, "graftDiagnosticSet: tiles per ticket %d", value);
            gSettings.tilesPerTicket = value;
            return GRAFT_OK;
        case GRAFT_DIAG_RESERVED_CUS:
            if (value < 0 || value > 128) return fail(GRAFT_ERR_ARG, "graftDiagnosticSet: %d reserved CUs", value);
            gSettings.reservedCUs = value;
            return GRAFT_OK;
        case GRAFT_DIAG_COPY_PIECE_KB:
            if (value < -1) return fail(GRAFT_ERR_ARG, "graftDiagnosticSet: copy pieces of %d KiB", value);
            gSettings.copyPieceKB = value;
            return GRAFT_OK;
        case GRAFT_DIAG_EMULATED_DEVICES:
            if (value < 0 || value > kMaxDevices) return fail(GRAFT_ERR_ARG, "graftDiagnosticSet: %d emulated devices", value);
            gEmulatedDevices = value;
            return GRAFT_OK;
        default:
            return fail(GRAFT_ERR_ARG, "graftDiagnosticSet: unknown setting %d", which);
    }
}

GRAFT_API int graftDiagnosticGet(int which) {
    readSettingsOnce();
    switch (which) {
        case GRAFT_DIAG_SCREEN_CAP: return gSettings.screenCap;
        case GRAFT_DIAG_GROUP_SIZE: return gSettings.groupSize;
        case GRAFT_DIAG_TILES_PER_TICKET: return gSettings.tilesPerTicket;
        case GRAFT_DIAG_RESERVED_CUS: return gSettings.reservedCUs;
        case GRAFT_DIAG_COPY_PIECE_KB: return gSettings.copyPieceKB;
        case GRAFT_DIAG_EMULATED_DEVICES: return gEmulatedDevices;
        case GRAFT_DIAG_TABLE_BUILDS: return (int)(gTables.builds & 0x7fffffff);
        default: return -1;
    }
}

// row iy of a host frame as 8-bit RGB triples (the quantiser of the RGB bar; NaN -> 0)
static void quantiseRow(const void *frameHost, int format, int nCols, int iy, unsigned char *row) {
    auto q = [](double v) { return (unsigned char)((v < 0 ? 0 : (v > 1 ? 1 : (v == v ? v : 0))) * 255 + 0.5); };
    for (int ix = 0; ix < nCols; ++ix) {
        const size_t p = (size_t)iy * nCols + ix;
        for (int c = 0; c < 3; ++c) {
            if (format == GRAFT_OUT_F64_RGB) row[3 * ix + c] = q(((const double *)frameHost)[3 * p + c]);
            else if (format == GRAFT_OUT_F32_RGBA) row[3 * ix + c] = q(((const float *)frameHost)[4 * p + c]);
            else row[3 * ix + c] = ((const unsigned char *)frameHost)[4 * p + c];
        }
    }
}

GRAFT_API int graftWritePPM(const char *path, const void *frameHost, int format, int nCols, int nRows) {
    if (!path || !frameHost || nCols <= 0 || nRows <= 0 || graftPixelBytes(format) == 0)
        return fail(GRAFT_ERR_ARG, "graftWritePPM: bad argument");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(GRAFT_ERR_ARG, "graftWritePPM: cannot open %s", path);
    fprintf(f, "P6\n%d %d\n255\n", nCols, nRows);
    std::vector<unsigned char> row(3 * (size_t)nCols);
    for (int iy = nRows - 1; iy >= 0; --iy) {
        quantiseRow(frameHost, format, nCols, iy, row.data());
        if (fwrite(row.data(), 1, row.size(), f) != row.size()) {
            fclose(f);
            return fail(GRAFT_ERR_ARG, "graftWritePPM: short write to %s", path);
        }
    }
    fclose(f);
    return GRAFT_OK;
}

// PNG: 8-bit RGB, no interlace, every scanline with filter type 0, the zlib stream made of STORED deflate blocks (no
// compression, no dependency): signature, IHDR, one IDAT, IEND; CRC-32 per chunk, Adler-32 over the scanlines.
GRAFT_API int graftWritePNG(const char *path, const void *frameHost, int format, int nCols, int nRows) {
    if (!path || !frameHost || nCols <= 0 || nRows <= 0 || graftPixelBytes(format) == 0)
        return fail(GRAFT_ERR_ARG, "graftWritePNG: bad argument");
    const unsigned long long lineBytes = 1ull + 3ull * (unsigned long long)nCols, raw = lineBytes * (unsigned long long)nRows;
    const unsigned long long blocks = (raw + 65534) / 65535, idat = 2 + raw + 5 * blocks + 4;
    if (idat > 0x7fffffffull) return fail(GRAFT_ERR_UNSUPPORTED, "graftWritePNG: a %d x %d frame does not fit one IDAT chunk (use graftWritePPM)", nCols, nRows);
    FILE *f = fopen(path, "wb");
    if (!f) return fail(GRAFT_ERR_ARG, "graftWritePNG: cannot open %s", path);
    static unsigned table[256];
    if (!table[1])
        for (unsigned n = 0; n < 256; ++n) {
            unsigned c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            table[n] = c;
        }
    bool ok = true;
    unsigned crc = 0;
    auto put = [&](const unsigned char *b, size_t n) {  // bytes of the chunk being written (type and data go into its CRC)
        for (size_t i = 0; i < n; ++i) crc = table[(crc ^ b[i]) & 0xff] ^ (crc >> 8);
        ok = ok && fwrite(b, 1, n, f) == n;
    };
    auto put32 = [&](unsigned v) { const unsigned char b[4] = {(unsigned char)(v >> 24), (unsigned char)(v >> 16), (unsigned char)(v >> 8), (unsigned char)v}; put(b, 4); };
    auto begin = [&](unsigned length, const char *type) { put32(length); crc = 0xffffffffu; put((const unsigned char *)type, 4); };
    auto end = [&]() { const unsigned c = crc ^ 0xffffffffu; put32(c); };
    const unsigned char signature[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    put(signature, 8);
    begin(13, "IHDR");
    put32((unsigned)nCols); put32((unsigned)nRows);
    const unsigned char header[5] = {8, 2, 0, 0, 0};  // bit depth 8, colour type 2 (RGB), deflate, filter method 0, no interlace
    put(header, 5);
    end();
    begin((unsigned)idat, "IDAT");
    const unsigned char zlibHeader[2] = {0x78, 0x01};
    put(zlibHeader, 2);
    std::vector<unsigned char> line((size_t)lineBytes);
    unsigned a = 1, b = 0;  // Adler-32
    unsigned long long sent = 0;  // bytes of scanline data written so far
    for (int iy = nRows - 1; iy >= 0 && ok; --iy) {  // top row first
        line[0] = 0;
        quantiseRow(frameHost, format, nCols, iy, line.data() + 1);
        for (size_t i = 0; i < line.size(); ++i) { a = (a + line[i]) % 65521u; b = (b + a) % 65521u; }
        size_t at = 0;
        while (at < line.size()) {
            if (sent % 65535 == 0) {  // a new stored block starts here
                const unsigned long long left = raw - sent;
                const unsigned len = (unsigned)(left < 65535 ? left : 65535);
                const unsigned char head[5] = {(unsigned char)(left <= 65535 ? 1 : 0), (unsigned char)len, (unsigned char)(len >> 8),
                                               (unsigned char)~len, (unsigned char)(~len >> 8)};
                put(head, 5);
            }
            const size_t room = (size_t)(65535 - sent % 65535), n = line.size() - at < room ? line.size() - at : room;
            put(line.data() + at, n);
            at += n;
            sent += n;
        }
    }
    put32((b << 16) | a);
    end();
    begin(0, "IEND");
    end();
    ok = fclose(f) == 0 && ok;
    if (!ok) return fail(GRAFT_ERR_ARG, "graftWritePNG: short write to %s", path);
    return GRAFT_OK;
}

GRAFT_API size_t graftPixelBytes(int format) {
    switch (format) {
        case GRAFT_OUT_F64_RGB: return 24;
        case GRAFT_OUT_F32_RGBA: return 16;
        case GRAFT_OUT_U8_RGBA: return 4;
        default: return 0;
    }
}

// ---- host-side steps of launchRays ----

GRAFT_API int graftLoopOffsets(double extent, int n, double *out, int cap) { return loopOffsets(extent, n, out, cap); }

GRAFT_API int graftScreenCoord(double coord, int n, double extent) { return screenCoord(coord, n, extent); }

GRAFT_API void graftCamInitialize(const double target[3], const double viewDir[3], double targetToScreen,
                                  double screenToCam, double screenCenter[3], double camPos[3]) {
    camInitializeHost(target, viewDir, targetToScreen, screenToCam, screenCenter, camPos);
}

// ---- pinned host memory ----

GRAFT_API int graftHostAlloc(size_t bytes, void **out) {
    if (!out || bytes == 0) return fail(GRAFT_ERR_ARG, "graftHostAlloc: null pointer or zero size");
    *out = nullptr;
    if (!gBound) return fail(GRAFT_ERR_STATE, "graftHostAlloc: call graftInit first");
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return GRAFT_OK;
}

GRAFT_API void graftHostFree(void *p) {
    if (p) (void)hipHostFree(p);
}

// ---- jobs ----

GRAFT_API int graftJobCreate(const GraftScene *scene, const GraftView *view, GraftJob **jobOut) {
    if (!jobOut) return fail(GRAFT_ERR_ARG, "graftJobCreate: null job pointer");
    *jobOut = nullptr;
    if (!gBound) return fail(GRAFT_ERR_STATE, "graftJobCreate: call graftInit first");
    PackedJob J;
    std::string err;
    const int rc = packJob(scene, view, gSettings, gTables, J, err);
    if (rc != GRAFT_OK) return fail(rc, "graftJobCreate: %s", err.c_str());
    GraftJob *job = new (std::nothrow) GraftJob();
    if (!job) return fail(GRAFT_ERR_HIP, "graftJobCreate: out of host memory");
    void *arena = nullptr;
    hipError_t e = hipMalloc(&arena, J.image.size());
    if (e != hipSuccess) {
        delete job;
        return fail(GRAFT_ERR_HIP, "hipMalloc(%zu): %s", J.image.size(), hipGetErrorString(e));
    }
    e = hipMemcpy(arena, J.image.data(), J.image.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(arena);
        delete job;
        return fail(GRAFT_ERR_HIP, "hipMemcpy(scene): %s", hipGetErrorString(e));
    }
    describe(job, J, arena, true, gBound);
    *jobOut = job;
    return GRAFT_OK;
}

GRAFT_API int graftJobScreenLevel(const GraftJob *job) { return job ? job->screen : -1; }

GRAFT_API void graftJobDestroy(GraftJob *job) {
    if (!job) return;
    if (job->dev && job->dev->stream) (void)hipStreamSynchronize(job->dev->stream);
    if (job->ownsArena) (void)hipFree(job->arena);
    delete job;
}

GRAFT_API int graftJobTrace(GraftJob *job, int row0, int nRows, int format, void *frame, int32_t *hitIndex,
                            GraftCounters *counters, void *stream) {
    if (!gBound) return fail(GRAFT_ERR_STATE, "graftJobTrace: call graftInit first");
    const int rc = checkTraceArgs("graftJobTrace", job, row0, nRows, format, frame);
    if (rc != GRAFT_OK) return rc;
    return launchTrace(job, row0, nRows, format, frame, hitIndex, counters, stream ? (hipStream_t)stream : job->dev->stream);
}

GRAFT_API int graftSynchronize(void) {
    if (!gBound) return fail(GRAFT_ERR_STATE, "graftSynchronize: call graftInit first");
    HIP_TRY(hipStreamSynchronize(gBound->stream));
    return GRAFT_OK;
}

// The job graftRenderRows traces: the one of the previous call when scene, view and settings are byte for byte the
// same (nothing is packed or uploaded then), otherwise a new one packed into the same device block.
static int cachedJob(Device *D, const GraftScene *scene, const GraftView *view, GraftJob **out) {
    if (!scene || !view) return fail(GRAFT_ERR_ARG, "graftRenderRows: null scene or view");
    if (view->nCols <= 0 || view->nRows <= 0 || !view->colOffsets || !view->rowOffsets)
        return fail(GRAFT_ERR_ARG, "graftRenderRows: bad frame %d x %d", view->nCols, view->nRows);
    std::vector<unsigned char> key;
    jobKey(scene, view, gSettings, key);
    if (D->cacheJob && key == D->cacheKey) {
        *out = D->cacheJob;
        return GRAFT_OK;
    }
    PackedJob J;
    std::string err;
    const int rc = packJob(scene, view, gSettings, gTables, J, err);
    if (rc != GRAFT_OK) return fail(rc, "graftJobCreate: %s", err.c_str());
    D->cacheKey.clear();  // whatever happens below, the old job's block is about to be overwritten
    hipError_t e = D->arena.reserve(J.image.size());
    if (e != hipSuccess) return fail(GRAFT_ERR_HIP, "hipMalloc(%zu): %s", J.image.size(), hipGetErrorString(e));
    HIP_TRY(hipMemcpy(D->arena.ptr, J.image.data(), J.image.size(), hipMemcpyHostToDevice));
    if (!D->cacheJob) D->cacheJob = new (std::nothrow) GraftJob();
    if (!D->cacheJob) return fail(GRAFT_ERR_HIP, "graftRenderRows: out of host memory");
    describe(D->cacheJob, J, D->arena.ptr, false, D);
    D->cacheKey.swap(key);
    *out = D->cacheJob;
    return GRAFT_OK;
}

GRAFT_API int graftRenderRows(const GraftScene *scene, const GraftView *view, int row0, int nRows, int format,
                              void *frameHost, int32_t *hitIndexHost, GraftCounters *countersHost) {
    if (!frameHost) return fail(GRAFT_ERR_ARG, "graftRenderRows: null frame");
    if (!gBound) return fail(GRAFT_ERR_STATE, "graftJobCreate: call graftInit first");
    Device *D = gBound;
    GraftJob *job = nullptr;
    int rc = cachedJob(D, scene, view, &job);
    if (rc != GRAFT_OK) return rc;
    const size_t px = (size_t)(nRows > 0 ? nRows : 0) * job->P.nCols;
    const size_t frameBytes = px * graftPixelBytes(format);
    if (frameBytes == 0) return fail(GRAFT_ERR_ARG, "graftRenderRows: empty band or unknown format");
    rc = checkTraceArgs("graftRenderRows", job, row0, nRows, format, frameHost);
    if (rc != GRAFT_OK) return rc;
    hipError_t e;
    if ((e = D->frame.reserve(frameBytes)) != hipSuccess) return fail(GRAFT_ERR_HIP, "hipMalloc(frame, %zu): %s", frameBytes, hipGetErrorString(e));
    int32_t *dIdx = nullptr;
    GraftCounters *dCnt = nullptr;
    if (hitIndexHost) {
        if ((e = D->index.reserve(px * 4)) != hipSuccess) return fail(GRAFT_ERR_HIP, "hipMalloc(index): %s", hipGetErrorString(e));
        dIdx = (int32_t *)D->index.ptr;
    }
    if (countersHost) {
        if ((e = D->counters.reserve(sizeof(GraftCounters))) != hipSuccess) return fail(GRAFT_ERR_HIP, "hipMalloc(counters): %s", hipGetErrorString(e));
        HIP_TRY(hipMemsetAsync(D->counters.ptr, 0, sizeof(GraftCounters), D->stream));
        dCnt = (GraftCounters *)D->counters.ptr;
    }
    // A large frame is traced in pieces of rows, and every finished piece leaves for the host (on the copy stream,
    // behind an event) while the next ones are traced: the copy of a 16384 x 16384 frame takes about as long as its
    // trace, and one after the other they would double the call.  How many pieces: every piece is a launch of its own
    // (measured: about 0.8 ms of a 16384-wide frame each) and the copy of the last one is not hidden, so a frame whose
    // copy is short beside its trace wants about sqrt(copy time / 0.8 ms) pieces -- sqrt(bytes / 48 MiB) at the 55 GB/s
    // of the link --, while a frame whose copy outlasts its trace wants the copy to start as early as possible: at
    // least one piece per 256 MiB.  At most 16; small frames are one piece, whose launch and copy are as before.
    // copyPieceKB > 0 forces pieces of about that size, 0 one piece.
    const int nCols = job->P.nCols;
    int pieces = 1;
    if (gSettings.copyPieceKB > 0) {
        const size_t want = frameBytes / ((size_t)gSettings.copyPieceKB << 10);
        pieces = (int)(want < 1 ? 1 : want > 16 ? 16 : want);
    } else if (gSettings.copyPieceKB < 0) {
        const double few = sqrt((double)frameBytes / (double)((size_t)48 << 20));
        const size_t many = frameBytes >> 28;
        const size_t want = (size_t)(few + 0.5) > many ? (size_t)(few + 0.5) : many;
        pieces = (int)(want < 1 ? 1 : want > 16 ? 16 : want);
    }
    const int tileRows = (int)(((long long)nRows + kTile - 1) / kTile);
    if (pieces > tileRows) pieces = tileRows;
    const size_t pixelBytes = graftPixelBytes(format);
    for (int k = 0; k < pieces && rc == GRAFT_OK; ++k) {
        const int r0 = (int)((long long)k * tileRows / pieces) * kTile;
        int r1 = (int)((long long)(k + 1) * tileRows / pieces) * kTile;
        if (r1 > nRows) r1 = nRows;
        if (r1 <= r0) continue;
        const size_t firstPx = (size_t)r0 * nCols, px1 = (size_t)(r1 - r0) * nCols;
        unsigned char *dFrame = (unsigned char *)D->frame.ptr + firstPx * pixelBytes;
        rc = launchTrace(job, row0 + r0, r1 - r0, format, dFrame, dIdx ? dIdx + firstPx : nullptr, dCnt, D->stream);
        if (rc != GRAFT_OK) break;
        hipStream_t out = D->stream;
        if (pieces > 1) {  // the copy waits for this piece's kernel only
            out = D->copyStream;
            if ((e = hipEventRecord(D->pieceTraced[k], D->stream)) != hipSuccess ||
                (e = hipStreamWaitEvent(out, D->pieceTraced[k], 0)) != hipSuccess) {
                rc = fail(GRAFT_ERR_HIP, "graftRenderRows: ordering a copy behind its piece: %s", hipGetErrorString(e));
                break;
            }
        }
        if ((e = hipMemcpyAsync((unsigned char *)frameHost + firstPx * pixelBytes, dFrame, px1 * pixelBytes, hipMemcpyDeviceToHost, out)) != hipSuccess)
            rc = fail(GRAFT_ERR_HIP, "hipMemcpy(frame): %s", hipGetErrorString(e));
        if (rc == GRAFT_OK && dIdx && (e = hipMemcpyAsync(hitIndexHost + firstPx, dIdx + firstPx, px1 * 4, hipMemcpyDeviceToHost, out)) != hipSuccess)
            rc = fail(GRAFT_ERR_HIP, "hipMemcpy(index): %s", hipGetErrorString(e));
    }
    if (rc == GRAFT_OK && dCnt && (e = hipMemcpyAsync(countersHost, dCnt, sizeof(GraftCounters), hipMemcpyDeviceToHost, D->stream)) != hipSuccess) rc = fail(GRAFT_ERR_HIP, "hipMemcpy(counters): %s", hipGetErrorString(e));
    if ((e = hipStreamSynchronize(D->stream)) != hipSuccess && rc == GRAFT_OK) rc = fail(GRAFT_ERR_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e));
    if (pieces > 1 && (e = hipStreamSynchronize(D->copyStream)) != hipSuccess && rc == GRAFT_OK) rc = fail(GRAFT_ERR_HIP, "hipStreamSynchronize(copies): %s", hipGetErrorString(e));
    return rc;
}

GRAFT_API int graftRenderRowsNoAlloc(const GraftScene *scene, const GraftView *view, int row0, int nRows, int format,
                                     void *frameDevice, int32_t *hitIndexDevice, GraftCounters *countersDevice,
                                     void *stream) {
    if (!gBound) return fail(GRAFT_ERR_STATE, "graftRenderRowsNoAlloc: call graftInit first");
    Device *D = gBound;
    PackedJob J;
    std::string err;
    int rc = packJob(scene, view, gSettings, gTables, J, err);
    if (rc != GRAFT_OK) return fail(rc, "graftRenderRowsNoAlloc: %s", err.c_str());
    if (J.image.size() > kScratchBytes)
        return fail(GRAFT_ERR_UNSUPPORTED, "graftRenderRowsNoAlloc: scene and view pack to %zu bytes, the static "
                    "scratch block holds %zu (use graftRenderRows or graftJobCreate)", J.image.size(), kScratchBytes);
    GraftJob job;
    describe(&job, J, D->scratch, false, D);
    rc = checkTraceArgs("graftRenderRowsNoAlloc", &job, row0, nRows, format, frameDevice);
    if (rc != GRAFT_OK) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : D->stream;
    // the staging block may still be the source of the previous call's upload
    HIP_TRY(hipEventSynchronize(D->scratchCopied));
    memcpy(D->scratchStage, J.image.data(), J.image.size());
    HIP_TRY(hipMemcpyAsync(D->scratch, D->scratchStage, J.image.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(D->scratchCopied, s));
    // (the ticket counters inside the block are set by the launch itself, behind the upload on the same stream)
    return launchTrace(&job, row0, nRows, format, frameDevice, hitIndexDevice, countersDevice, s);
}

// ---- several GPUs driven by one process ----

GRAFT_API int graftFramePlan(int nRows, int nDevices, int piecesPerDevice, int *firstRow, int *rows, int *owner,
                             int *rowInOwner, int *ownedRows, int cap) {
    return framePlan(nRows, nDevices, piecesPerDevice, firstRow, rows, owner, rowInOwner, ownedRows, cap);
}

GRAFT_API int graftFramePieces(int nRows, int nDevices, int piecesPerDevice, int *firstRow, int *rows, int *owner, int cap) {
    return graftFramePlan(nRows, nDevices, piecesPerDevice, firstRow, rows, owner, nullptr, nullptr, cap);
}

GRAFT_API int graftInitDevices(int nDevices) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    const bool emulate = nDevices > n && nDevices <= gEmulatedDevices && n >= 1;
    if (nDevices < 1 || (nDevices > n && !emulate) || nDevices > kMaxDevices)
        return fail(GRAFT_ERR_ARG, "graftInitDevices: %d devices asked for, %d present (at most %d)", nDevices, n, kMaxDevices);
    if (gBound && gBound->id != 0) return fail(GRAFT_ERR_STATE, "graftInitDevices: already bound to device %d (device 0 is the root)", gBound->id);
    if (gMultiDevices >= nDevices) {  // the first nDevices of the devices bound already (ranks 0 .. n-1 of their communicator exchange among themselves)
        gMultiDevices = nDevices;
        return GRAFT_OK;
    }
    // Until every step below has succeeded no multi-device frame can be started: a failure leaves the library with
    // device 0 only (gMultiDevices = 0: graftRenderFrameMulti refuses), never with half a set of communicators.
    const int before = gMultiDevices;
    gMultiDevices = 0;
    for (int d = 0; d < nDevices; ++d) {
        Device *D = nullptr;
        const int rc = initDevice(d, &D, emulate ? d % n : d);
        if (rc != GRAFT_OK) return rc;
        HIP_TRY(hipSetDevice(D->physical));
        if (!D->moveStream) HIP_TRY(hipStreamCreateWithFlags(&D->moveStream, hipStreamNonBlocking));
        if (!D->traced) HIP_TRY(hipEventCreateWithFlags(&D->traced, hipEventDisableTiming));
        if (!D->frameStart) HIP_TRY(hipEventCreate(&D->frameStart));
        if (!D->kernelsDone) HIP_TRY(hipEventCreate(&D->kernelsDone));
        if (!D->movesDone) HIP_TRY(hipEventCreate(&D->movesDone));
    }
    if (nDevices > 1 && !emulate) {
        if (!gRccl.handle) {
            void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
            if (!h) return fail(GRAFT_ERR_UNSUPPORTED, "graftInitDevices: cannot load librccl.so: %s", dlerror());
            gRccl.commInitAll = (decltype(gRccl.commInitAll))dlsym(h, "ncclCommInitAll");
            gRccl.commDestroy = (decltype(gRccl.commDestroy))dlsym(h, "ncclCommDestroy");
            gRccl.groupStart = (decltype(gRccl.groupStart))dlsym(h, "ncclGroupStart");
            gRccl.groupEnd = (decltype(gRccl.groupEnd))dlsym(h, "ncclGroupEnd");
            gRccl.send = (decltype(gRccl.send))dlsym(h, "ncclSend");
            gRccl.recv = (decltype(gRccl.recv))dlsym(h, "ncclRecv");
            gRccl.errorString = (decltype(gRccl.errorString))dlsym(h, "ncclGetErrorString");
            if (!gRccl.commInitAll || !gRccl.commDestroy || !gRccl.groupStart || !gRccl.groupEnd || !gRccl.send ||
                !gRccl.recv || !gRccl.errorString) {
                dlclose(h);
                return fail(GRAFT_ERR_UNSUPPORTED, "graftInitDevices: librccl.so lacks an entry point");
            }
            gRccl.handle = h;
        }
        for (int d = 0; d < before; ++d)  // a smaller communicator from an earlier call
            if (gDevices[d].comm) { (void)gRccl.commDestroy(gDevices[d].comm); gDevices[d].comm = nullptr; }
        ncclComm_t comms[kMaxDevices];
        int ids[kMaxDevices];
        for (int d = 0; d < nDevices; ++d) ids[d] = d;
        const ncclResult_t r = gRccl.commInitAll(comms, nDevices, ids);
        if (r != ncclSuccess) {
            (void)hipSetDevice(0);
            return fail(GRAFT_ERR_HIP, "ncclCommInitAll(%d): %s", nDevices, gRccl.errorString(r));
        }
        for (int d = 0; d < nDevices; ++d) gDevices[d].comm = comms[d];
    }
    HIP_TRY(hipSetDevice(0));
    gBound = &gDevices[0];
    gMultiDevices = nDevices;
    gMultiEmulated = emulate;
    return GRAFT_OK;
}

// The job every one of devices 0 .. G-1 traces: packed ONCE on the host; a device whose cached job was packed from
// the same input keeps it, the others receive the image.
static int cachedJobs(int G, const GraftScene *scene, const GraftView *view, GraftJob **jobs) {
    if (!scene || !view) return fail(GRAFT_ERR_ARG, "graftRenderFrameMulti: null scene or view");
    if (view->nCols <= 0 || view->nRows <= 0 || !view->colOffsets || !view->rowOffsets)
        return fail(GRAFT_ERR_ARG, "graftRenderFrameMulti: bad frame %d x %d", view->nCols, view->nRows);
    std::vector<unsigned char> key;
    jobKey(scene, view, gSettings, key);
    PackedJob J;
    bool packed = false;
    for (int d = 0; d < G; ++d) {
        Device *D = &gDevices[d];
        if (D->cacheJob && key == D->cacheKey) {
            jobs[d] = D->cacheJob;
            continue;
        }
        if (!packed) {
            std::string err;
            const int rc = packJob(scene, view, gSettings, gTables, J, err);
            if (rc != GRAFT_OK) return fail(rc, "graftRenderFrameMulti: %s", err.c_str());
            packed = true;
        }
        HIP_TRY(hipSetDevice(D->physical));
        D->cacheKey.clear();  // whatever happens below, the old job's block is about to be overwritten
        hipError_t e = D->arena.reserve(J.image.size());
        if (e != hipSuccess) return fail(GRAFT_ERR_HIP, "hipMalloc(%zu) on device %d: %s", J.image.size(), d, hipGetErrorString(e));
        HIP_TRY(hipMemcpy(D->arena.ptr, J.image.data(), J.image.size(), hipMemcpyHostToDevice));
        if (!D->cacheJob) D->cacheJob = new (std::nothrow) GraftJob();
        if (!D->cacheJob) return fail(GRAFT_ERR_HIP, "graftRenderFrameMulti: out of host memory");
        describe(D->cacheJob, J, D->arena.ptr, false, D);
        D->cacheKey = key;
        jobs[d] = D->cacheJob;
    }
    return GRAFT_OK;
}

#define NCCL_TRY(expr)                                                                                    \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess) { rc = fail(GRAFT_ERR_HIP, "%s: %s", #expr, gRccl.errorString(r_)); goto done; } \
    } while (0)
#define HIP_GO(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) { rc = fail(GRAFT_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); goto done; } \
    } while (0)

GRAFT_API int graftRenderFrameMulti(const GraftScene *scene, const GraftView *view, int format, int piecesPerDevice,
                                    void *frameHost, int32_t *hitIndexHost, GraftCounters *countersHost) {
    if (!frameHost) return fail(GRAFT_ERR_ARG, "graftRenderFrameMulti: null frame");
    if (gMultiDevices < 1) return fail(GRAFT_ERR_STATE, "graftRenderFrameMulti: call graftInitDevices first");
    const size_t pixelBytes = graftPixelBytes(format);
    if (pixelBytes == 0) return fail(GRAFT_ERR_ARG, "graftRenderFrameMulti: unknown format %d", format);
    const int G = gMultiDevices;
    if (G == 1) {  // one GPU: exactly graftRenderRows
        if (!view) return fail(GRAFT_ERR_ARG, "graftRenderFrameMulti: null view");
        return graftRenderRows(scene, view, 0, view->nRows, format, frameHost, hitIndexHost, countersHost);
    }
    const bool emulated = gMultiEmulated;
    for (int d = 0; d < G; ++d)
        if ((!emulated && !gDevices[d].comm) || !gDevices[d].moveStream)
            return fail(GRAFT_ERR_STATE, "graftRenderFrameMulti: device %d has no communicator (graftInitDevices failed?)", d);
    int rc = GRAFT_OK;
    bool groupOpen = false;
    GraftJob *jobs[kMaxDevices] = {nullptr};
    std::vector<int> first, count, owner, at;  // `at`: a piece's first row inside its owner's buffer
    int ownedRows[kMaxDevices] = {0};
    GraftCounters sum = {0, 0, 0, 0};
    Device &root = gDevices[0];
    int nPieces = 0, rounds = 0, nCols = 0, nRows = 0;
    size_t hostRows = 0;
    // A trace that shares its GPU with RCCL's send/receive kernels leaves them a few CUs: without that the transfer of
    // round c could only start when a persistent workgroup of round c + 1's trace retires, i.e. near its end.
    const int reservedBefore = gSettings.reservedCUs;
    if (gSettings.reservedCUs < kMultiReservedCUs) gSettings.reservedCUs = kMultiReservedCUs;
    if ((rc = cachedJobs(G, scene, view, jobs)) != GRAFT_OK) goto done;
    nCols = jobs[0]->P.nCols;
    nRows = jobs[0]->P.nRows;
    nPieces = graftFramePlan(nRows, G, piecesPerDevice, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
    first.resize(nPieces); count.resize(nPieces); owner.resize(nPieces); at.resize(nPieces);
    graftFramePlan(nRows, G, piecesPerDevice, first.data(), count.data(), owner.data(), at.data(), ownedRows, nPieces);
    // buffers: the root holds the whole frame, every other device its own pieces
    for (int d = 0; d < G; ++d) {
        Device &D = gDevices[d];
        HIP_GO(hipSetDevice(D.physical));
        const size_t rowsHere = d == 0 ? (size_t)nRows : (size_t)ownedRows[d];
        HIP_GO((d == 0 ? D.frame : D.pieces).reserve(rowsHere * nCols * pixelBytes + 16));
        if (hitIndexHost) HIP_GO((d == 0 ? D.index : D.pieceIndex).reserve(rowsHere * nCols * 4 + 16));
        if (countersHost) {
            HIP_GO(D.counters.reserve(sizeof(GraftCounters)));
            HIP_GO(hipMemsetAsync(D.counters.ptr, 0, sizeof(GraftCounters), D.stream));
        }
        HIP_GO(hipEventRecord(D.frameStart, D.stream));
        HIP_GO(hipEventRecord(D.kernelsDone, D.stream));  // (re-recorded behind every kernel)
    }
    // Round c: every device queues the kernel of its c-th piece; then one group of transfers moves the pieces of round
    // c to the root (each sender's transfer is ordered behind its kernel through its `traced` event and runs on its
    // move stream, the root receives on its move stream), while the kernels of round c + 1 are queued behind them on
    // the trace streams and leave the transfer kernels their reserved CUs.
    rounds = (nPieces + G - 1) / G;
    for (int c = 0; c < rounds; ++c) {
        for (int k = c * G; k < nPieces && k < (c + 1) * G; ++k) {
            Device &D = gDevices[owner[k]];
            HIP_GO(hipSetDevice(D.physical));
            unsigned char *frame = owner[k] == 0 ? (unsigned char *)D.frame.ptr + (size_t)first[k] * nCols * pixelBytes
                                                 : (unsigned char *)D.pieces.ptr + (size_t)at[k] * nCols * pixelBytes;
            int32_t *index = !hitIndexHost ? nullptr
                           : owner[k] == 0 ? (int32_t *)D.index.ptr + (size_t)first[k] * nCols
                                           : (int32_t *)D.pieceIndex.ptr + (size_t)at[k] * nCols;
            if ((rc = launchTrace(jobs[owner[k]], first[k], count[k], format, frame, index,
                                  countersHost ? (GraftCounters *)D.counters.ptr : nullptr, D.stream)) != GRAFT_OK) goto done;
            HIP_GO(hipEventRecord(D.kernelsDone, D.stream));
            if (owner[k] != 0) {
                HIP_GO(hipEventRecord(D.traced, D.stream));
                HIP_GO(hipStreamWaitEvent(D.moveStream, D.traced, 0));
            } else if (c < kHostCopyRounds) {
                HIP_GO(hipEventRecord(D.pieceTraced[c], D.stream));  // the root's own piece of this round is in its frame
            }
        }
        if (!emulated) {
            NCCL_TRY(gRccl.groupStart());
            groupOpen = true;
        }
        for (int k = c * G; k < nPieces && k < (c + 1) * G; ++k) {
            if (owner[k] == 0) continue;
            Device &D = gDevices[owner[k]];
            const size_t bytes = (size_t)count[k] * nCols * pixelBytes, ib = (size_t)count[k] * nCols * 4;
            unsigned char *src = (unsigned char *)D.pieces.ptr + (size_t)at[k] * nCols * pixelBytes;
            unsigned char *dst = (unsigned char *)root.frame.ptr + (size_t)first[k] * nCols * pixelBytes;
            int32_t *srcIndex = hitIndexHost ? (int32_t *)D.pieceIndex.ptr + (size_t)at[k] * nCols : nullptr;
            int32_t *dstIndex = hitIndexHost ? (int32_t *)root.index.ptr + (size_t)first[k] * nCols : nullptr;
            if (emulated) {  // the sender's transfer stream copies the piece; the root's transfer stream waits for it
                HIP_GO(hipSetDevice(D.physical));
                HIP_GO(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, D.moveStream));
                if (hitIndexHost) HIP_GO(hipMemcpyAsync(dstIndex, srcIndex, ib, hipMemcpyDeviceToDevice, D.moveStream));
                HIP_GO(hipEventRecord(D.traced, D.moveStream));  // (its wait on the kernel has been queued: the event is free again)
                HIP_GO(hipSetDevice(root.physical));
                HIP_GO(hipStreamWaitEvent(root.moveStream, D.traced, 0));
                continue;
            }
            NCCL_TRY(gRccl.send(src, bytes, ncclChar, 0, D.comm, D.moveStream));
            NCCL_TRY(gRccl.recv(dst, bytes, ncclChar, owner[k], root.comm, root.moveStream));
            if (hitIndexHost) {
                NCCL_TRY(gRccl.send(srcIndex, ib, ncclChar, 0, D.comm, D.moveStream));
                NCCL_TRY(gRccl.recv(dstIndex, ib, ncclChar, owner[k], root.comm, root.moveStream));
            }
        }
        if (!emulated) {
            groupOpen = false;
            NCCL_TRY(gRccl.groupEnd());
        }
        // The pieces of round c are neighbours in the frame (pieces are dealt in row order): once the root's own kernel and
        // the round's transfers are done, that block of rows leaves for the host on the root's copy stream, beside the
        // tracing and the transfers of the later rounds.
        if (c < kHostCopyRounds) {
            const int kFirst = c * G, kLast = (nPieces < (c + 1) * G ? nPieces : (c + 1) * G) - 1;
            const size_t row0 = (size_t)first[kFirst], rows = (size_t)(first[kLast] + count[kLast]) - row0;
            HIP_GO(hipSetDevice(root.physical));
            HIP_GO(hipEventRecord(root.roundMoved[c], root.moveStream));
            HIP_GO(hipStreamWaitEvent(root.copyStream, root.roundMoved[c], 0));
            if (owner[kFirst] == 0) HIP_GO(hipStreamWaitEvent(root.copyStream, root.pieceTraced[c], 0));
            HIP_GO(hipMemcpyAsync((unsigned char *)frameHost + row0 * nCols * pixelBytes, (unsigned char *)root.frame.ptr + row0 * nCols * pixelBytes,
                                  rows * nCols * pixelBytes, hipMemcpyDeviceToHost, root.copyStream));
            if (hitIndexHost)
                HIP_GO(hipMemcpyAsync(hitIndexHost + row0 * nCols, (int32_t *)root.index.ptr + row0 * nCols, rows * nCols * 4,
                                      hipMemcpyDeviceToHost, root.copyStream));
            hostRows = row0 + rows;  // rows [0, hostRows) are on their way to the host
        }
    }
    // the assembled frame leaves the root once its own kernels and every transfer are done
    for (int d = 0; d < G; ++d) {
        HIP_GO(hipSetDevice(gDevices[d].physical));
        HIP_GO(hipEventRecord(gDevices[d].movesDone, gDevices[d].moveStream));
        HIP_GO(hipStreamSynchronize(gDevices[d].stream));
        HIP_GO(hipStreamSynchronize(gDevices[d].moveStream));
        if (countersHost) {
            GraftCounters c = {0, 0, 0, 0};
            HIP_GO(hipMemcpy(&c, gDevices[d].counters.ptr, sizeof c, hipMemcpyDeviceToHost));
            sum.rays += c.rays;
            sum.tests += c.tests;
            sum.raysTraced += c.raysTraced;
        }
        // what graftMultiStats reports: this device's kernels from first to last, and how long after its last kernel
        // its last transfer ended (the part of the exchange that tracing did not hide; negative: none)
        HIP_GO(hipEventElapsedTime(&gMultiTraceMs[d], gDevices[d].frameStart, gDevices[d].kernelsDone));
        HIP_GO(hipEventElapsedTime(&gMultiTailMs[d], gDevices[d].kernelsDone, gDevices[d].movesDone));
    }
    gMultiStatsDevices = G;
    HIP_GO(hipSetDevice(0));
    HIP_GO(hipStreamSynchronize(root.copyStream));
    if (hostRows < (size_t)nRows) {  // (more rounds than copy events: the rest of the frame in one piece)
        HIP_GO(hipMemcpy((unsigned char *)frameHost + hostRows * nCols * pixelBytes, (unsigned char *)root.frame.ptr + hostRows * nCols * pixelBytes,
                         ((size_t)nRows - hostRows) * nCols * pixelBytes, hipMemcpyDeviceToHost));
        if (hitIndexHost)
            HIP_GO(hipMemcpy(hitIndexHost + hostRows * nCols, (int32_t *)root.index.ptr + hostRows * nCols, ((size_t)nRows - hostRows) * nCols * 4,
                             hipMemcpyDeviceToHost));
    }
    if (countersHost) *countersHost = sum;
done:
    if (groupOpen) (void)gRccl.groupEnd();  // never leave a group open behind an error
    if (rc != GRAFT_OK)
        for (int d = 0; d < G; ++d) {  // let what was queued drain before the caller sees the error
            (void)hipSetDevice(gDevices[d].physical);
            if (gDevices[d].stream) (void)hipStreamSynchronize(gDevices[d].stream);
            if (gDevices[d].moveStream) (void)hipStreamSynchronize(gDevices[d].moveStream);
            if (gDevices[d].copyStream) (void)hipStreamSynchronize(gDevices[d].copyStream);
        }
    gSettings.reservedCUs = reservedBefore;
    (void)hipSetDevice(0);
    return rc;
}

GRAFT_API int graftMultiStats(int device, float *traceMs, float *tailMs) {
    if (device < 0 || device >= gMultiStatsDevices) return fail(GRAFT_ERR_ARG, "graftMultiStats: device %d took part in no multi-device frame", device);
    if (traceMs) *traceMs = gMultiTraceMs[device];
    if (tailMs) *tailMs = gMultiTailMs[device];
    return GRAFT_OK;
}

static int probe(int which, const double *a, const double *b, double *out, size_t n) {
    if (!gBound) return fail(GRAFT_ERR_STATE, "probe: call graftInit first");
    if (!a || !out || (which && !b) || n == 0) return fail(GRAFT_ERR_ARG, "probe: null argument");
    hipStream_t stream = gBound->stream;
    double *dA = nullptr, *dB = nullptr, *dO = nullptr;
    auto run = [&]() -> int {
        HIP_TRY(hipMalloc((void **)&dA, n * 8));
        HIP_TRY(hipMalloc((void **)&dO, n * 8));
        HIP_TRY(hipMemcpy(dA, a, n * 8, hipMemcpyHostToDevice));
        if (which) {
            HIP_TRY(hipMalloc((void **)&dB, n * 8));
            HIP_TRY(hipMemcpy(dB, b, n * 8, hipMemcpyHostToDevice));
        }
        const unsigned grid = (unsigned)((n + 255) / 256);
        if (which) hipLaunchKernelGGL(probeDivKernel, dim3(grid), dim3(256), 0, stream, dA, dB, dO, n);
        else hipLaunchKernelGGL(probeSqrtKernel, dim3(grid), dim3(256), 0, stream, dA, dO, n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(out, dO, n * 8, hipMemcpyDeviceToHost));
        return GRAFT_OK;
    };
    const int rc = run();
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dO);  // on every path
    return rc;
}

GRAFT_API int graftProbeUnit(const double *v, double *out, size_t n) {
    if (!gBound) return fail(GRAFT_ERR_STATE, "probe: call graftInit first");
    if (!v || !out || n == 0) return fail(GRAFT_ERR_ARG, "probe: null argument");
    hipStream_t stream = gBound->stream;
    double *dV = nullptr, *dO = nullptr;
    auto run = [&]() -> int {
        HIP_TRY(hipMalloc((void **)&dV, n * 24));
        HIP_TRY(hipMalloc((void **)&dO, n * 24));
        HIP_TRY(hipMemcpy(dV, v, n * 24, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probeUnitKernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dV, dO, n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(out, dO, n * 24, hipMemcpyDeviceToHost));
        return GRAFT_OK;
    };
    const int rc = run();
    (void)hipFree(dV); (void)hipFree(dO);  // on every path
    return rc;
}

GRAFT_API int graftProbeSqrt(const double *a, double *out, size_t n) { return probe(0, a, nullptr, out, n); }
GRAFT_API int graftProbeDiv(const double *a, const double *b, double *out, size_t n) { return probe(1, a, b, out, n); }

}  // extern "C"
