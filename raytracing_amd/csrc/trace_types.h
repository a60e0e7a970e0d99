// trace_types.h -- constants and plain structs shared by the device code and the host-side packers of
// libgrafthip.so (included by grafthip.hip, and -- without the device-only declarations -- by the host-only
// sanitizer build of the packers, tests/host_asan).
#pragma once

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#endif

#include <cstddef>

#include "grafthip.h"

namespace graft {


constexpr int kTile = 8;            // 8x8 pixels = one wave64
constexpr int kMaxScreen = 7;       // highest screen level (TraceParams, traceKernel's SCREEN)
constexpr int kMaxDepth = 12;       // levels of the reflection chain kept per lane
constexpr int kSmallBlock = 256;    // scenes whose geometry fits a quarter of the LDS
#ifndef GRAFT_LARGE_BLOCK
#define GRAFT_LARGE_BLOCK 1024
#endif
constexpr int kLargeBlock = GRAFT_LARGE_BLOCK;   // one workgroup per CU sharing up to the whole LDS
constexpr int kGeomPad = 32;        // doubles of never-hit padding behind the packed geometry (the exact loop reads ahead)
constexpr int kSkipBehindMinShapes = 32;
constexpr int kCursors = 16;        // ticket counters of the many-counter kernels ...
constexpr int kCursorStride = 32;   // ... one per 128-byte line
constexpr size_t kLdsSmall = 36 * 1024;
constexpr size_t kLdsMax = 152 * 1024;
constexpr int kSampleTablePasses = 8;  // supersampling up to this many passes keeps per-tile scale factors in LDS
constexpr int kTileListCap = 128;   // entries of a wave's tile list (closestHitLists); longer lists fall back to all spheres

struct DevRun {  // shapes [firstShape, firstShape+count) are all of `kind`; their geometry starts at slot
    int kind, firstShape, count, slot;
    int fslot;    // sphere runs: first entry of the run's block in the single-precision screen table
    int fcount;   // entries of that block: count rounded up to a multiple of 8 with never-passing dummies
    int pad[2];
};

struct DevMaterial {  // 64 bytes per shape
    double color[3];
    double reflectivity, ambientLight, transparency, refrIndex;
    int slot, kind;
};

// What the shading of a hit WITHOUT children needs (scenes without mirrors or transparent shapes), formed on the host
// with the reference's own operations (mainTracing.c:352-369), which depend on the shape alone:
// wDiff = (1 - ambient)(1 - reflectivity)(1 - transparency), ambient[c] = ambientLight * color[c] + 0.0,
// mirrorZero = ((1 - ambient) * reflectivity) * 0.0 -- the reflection term of a hit that reflects nothing: a zero whose
// sign the sum still sees.
struct DevShade {  // 64 bytes per shape
    double color[3];
    double wDiff;
    double ambient[3];
    double mirrorZero;
};

struct DevLight {  // 64 bytes per light
    double color[3];
    double loc[3];
    int avoidFirst;  // the one avoided shape (avoid mode 1) or -1
    int avoidNum;
    int pad[2];
};

// Camera and background: needed once per sample, so they live in device memory and are fetched (scalar loads) where
// they are used; as kernel arguments they would occupy 44 scalar registers for the whole kernel and push the
// compiler into spilling scalars through vector lanes.
struct DevCamera {
    // mainTracing.c:54-62 after camInitialize
    double screenCenter[3], camPos[3], viewDir[3], lrVec[3], udVec[3];
    double lrDen, udDen;  // dir.dir of lrVec / udVec as getNewPoint forms it (mainTracing.c:107)
    double background[3];
    double subStepX, subStepY;
    // for the cone that bounds a tile's primary rays (closestHitLists): screenCenter - camPos and the unit vectors
    // along lrVec and udVec; the point of world offsets (x, y) lies at toScreen + x lrUnit + y udUnit from the camera
    double toScreen[3], lrUnit[3], udUnit[3];
};

// The same block seen as doubles in the constant address space: loads through this view are scalar loads whatever
// else the kernel stores to global memory (the block is written before the launch and never by a kernel).
#ifdef __HIPCC__
typedef const double __attribute__((address_space(4))) *ConstDoubles;
#endif
enum { CAM_SCREEN = 0, CAM_POS = 3, CAM_VIEW = 6, CAM_LR = 9, CAM_UD = 12, CAM_LRDEN = 15, CAM_UDDEN = 16, CAM_BACKGROUND = 17,
       CAM_SUBX = 20, CAM_SUBY = 21, CAM_TOSCREEN = 22, CAM_LRUNIT = 25, CAM_UDUNIT = 28 };
static_assert(offsetof(DevCamera, camPos) == 8 * CAM_POS && offsetof(DevCamera, udVec) == 8 * CAM_UD &&
              offsetof(DevCamera, lrDen) == 8 * CAM_LRDEN && offsetof(DevCamera, background) == 8 * CAM_BACKGROUND &&
              offsetof(DevCamera, subStepY) == 8 * CAM_SUBY && offsetof(DevCamera, udUnit) == 8 * CAM_UDUNIT,
              "DevCamera layout");

// Grid of closestHitGrid, read by scalar loads like the camera block: lo[3], hi[3], cell size[3], 1 / cell size[3].
struct DevGrid {
    double lo[3], hi[3], cs[3], invCs[3];
};
enum { GRID_LO = 0, GRID_HI = 3, GRID_CS = 6, GRID_INVCS = 9 };

struct TraceParams {
    const DevCamera *camera;
    int projectionType, aliasing, numPasses, maxDepth;
    int numShapes, numLights, numRuns, numSpheres;
    int nCols, nRows;          // whole frame
    int row0, bandRows;        // rows traced by this launch
    int tilesX, numTiles;
    int tilesPerTicket;        // consecutive tiles a wave takes per draw from a counter
    int prefetchTickets;       // draw the next ticket one ticket ahead (plenty of tickets per wave)
    int coarseTickets;         // one counter: tickets below this number hold tilesPerTicket tiles, the later ones one tile
    int geomDoubles;           // doubles of packed geometry (spheres 4 each, then planes 6 each)
    int avoidTotal;            // shapes on the lights' avoid lists, all lights together (ray-shape evaluations they spare)
    int skipLightsBehind;      // lights behind the shaded surface may go unasked: every colour, weight and colour product of
                               // the job is finite (0 * colour = 0) and the scene has kSkipBehindMinShapes shapes or more
    unsigned numPassesRecip;   // ceil(2^32 / numPasses): sample / numPasses = (sample * this) >> 32 for sample < 2^20
    int fusedMiss;             // no sphere has a squared radius below 1e-280: the exact loop may prove misses with FMAs
    int format;
    int filterFloats;          // floats of the single-precision screen table staged in LDS, 0 = no screen
    int groupEntries;          // grouped screen: entries of the group table (multiple of 8), 0 = not grouped
    int memberBase;            // grouped screen: first entry (float4 index) of the member table inside the screen table
    int groupSize;             // grouped screen: member entries per group (a multiple of 4)
    int exactLdsOffset;        // bytes from the start of LDS to the staged exact doubles (EXACTLDS kernels)
    // per-family screen (SCREEN == 3): `families` interleaved two-level tables; cone families first, ball table last
    int families, ballFamily, cameraFamily, lightFamilyBase;  // cameraFamily < 0: primary rays use the ball table
    double familyScale;        // rays whose |origin|_1 exceeds this use the exact loop
    // SCREEN == 4 (large scenes; three levels, cone member sections stay in global memory): LDS holds the first
    // ldsHeadFloats floats of geomF (all group sections), then ldsTailFloats floats copied from geomF + ldsTailSrc:
    // the ball family's member section (float4 index ballMembersLds) and the interleaved third-level sections
    // (float4 index superLds, superEntries entries per family, a multiple of 8)
    int ldsHeadFloats, ballMembersLds, ldsTailSrc, ldsTailFloats;
    int superLds, superEntries;
    // SCREEN == 5: level 4 with a uniform grid in place of the ball family (cells, entries and the per-sphere screen
    // table staged in LDS behind the third-level sections, at these byte offsets from the start of LDS)
    int gridNx, gridNy, gridNz;
    int gridCellsOff, gridRefsOff, gridAosOff;
    const DevGrid *grid;
    // SCREEN == 6, 7 (candidate lists, closestHitLists): geomF holds an image laid out as {sphere screen table
    // [numSpheres] float4 | level 6: exact doubles at exactLdsOffset | level 7 with mirrors: the uniform grid's cells
    // and entries (gridCellsOff, gridRefsOff; gridAosOff = the screen table) | camera cone table: listGroups group
    // entries as {axis, cos} float4 at listGroupsOff and their sines (float) at listSinesOff; then the listGroups *
    // groupSize member entries at listMembersOff, their sines at listMemberSinesOff and the members' sphere numbers
    // (16 bits) at listShapeOff | occluder-list index
    // [numSpheres][numLights] | list pool (16-bit sphere numbers; the list of all spheres first) | one tile list of
    // kTileListCap entries per wave of the workgroup | with supersampling of at most kSampleTablePasses passes: per
    // wave 16 * numPasses doubles, the scale factors of the tile's columns and rows (sampleTableOff, else -1)}; byte
    // offsets from the start of the image.  Level 6 stages the whole image in LDS; level 7 its first ldsHeadFloats
    // floats (everything up to and including the camera table's group entries, which every tile list scans), reads
    // the rest from global memory, and keeps the tile lists and sample tables in LDS behind the staged part
    int listGroups, listGroupsOff, listSinesOff, listMembersOff, listMemberSinesOff, listShapeOff, listIndexOff, listPoolOff, listTileOff, sampleTableOff;
    const int *memberShape;    // grouped screen: shape index of every member-table entry, -1 = padding
    const float *geomF;        // {cx, cy, cz, rho} per sphere in single precision, rho = inflated radius^2
    const double *geom;
    const DevRun *runs;
    const DevMaterial *materials;
    const DevShade *shade;
    const DevLight *lights;
    const unsigned char *avoidMask;  // [numLights][numShapes], avoid mode 2 only
    const double *colOff, *rowOff;
    void *frame;
    int *hitIndex;
    unsigned long long *counters;  // GraftCounters or null
    int *tileCursor;
};

#ifdef __HIPCC__
extern __shared__ __attribute__((aligned(32))) double ldsGeom[];

typedef double double4v __attribute__((ext_vector_type(4)));
#endif

enum { M_RADIANCE = 0, M_SHADOW = 1, M_INDEX = 2, M_DONE = 3 };
enum { F_MIRROR_PENDING = 1, F_HAS_REFRACTION = 2, F_REFRACTION_REFLECTS = 4 };

struct Hit {
    int idx, slot, kind;
    double x, y, z;
};


}  // namespace graft
