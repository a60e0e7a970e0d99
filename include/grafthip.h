/*
 * grafthip.h -- C ABI of libgrafthip.so: the per-pixel path of the ray tracer (ray generation, closest hit
 * over spheres and planes, shadow rays, diffuse + ambient shading, mirror-reflection chain) as hand-written
 * HIP kernels for AMD Instinct MI355X (gfx950).
 *
 * The library replaces the body of `void launchRays(void)` (reference mainTracing.c:387-443) and everything
 * it calls (mainTracing.c:104-372, shapes.c:38-101, vector.c:18-82).  The reference reads its inputs from
 * file-scope globals (mainTracing.c:29-67); here they travel in three plain structs.  The reference hands
 * its result to pixSetRGB(x, y, r, g, b) once per pixel (mainTracing.c:422, :439; 000pixel.h:55); here the
 * caller receives a frame indexed by the LOOP indices (iy, ix) of mainTracing.c:392-393 and feeds pixSetRGB
 * from it (raytracing_amd/csrc/launch_rays_dropin.c shows that loop).
 *
 * Conventions: every function returns GRAFT_OK (0) or a GRAFT_ERR_* code (the only convention the
 * reference documents is "0 if no error", 000pixel.h:30); graftLastError() describes the last failure on
 * the calling thread's process.  One process drives one GPU; the library is not re-entrant (neither is the
 * reference: launchRays is only ever called from the one event-loop thread, mainTracing.c:630, :642).
 * All arithmetic on the path is IEEE double in the reference's operation order.
 */
#ifndef GRAFTHIP_H
#define GRAFTHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRAFT_API __attribute__((visibility("default")))

enum { GRAFT_OK = 0, GRAFT_ERR_ARG = 1, GRAFT_ERR_UNSUPPORTED = 2, GRAFT_ERR_HIP = 3, GRAFT_ERR_STATE = 4 };

enum { GRAFT_SPHERE = 0, GRAFT_PLANE = 1 };             /* shape.intersection == sphereIntersect | planeIntersect */
enum { GRAFT_ORTHOGRAPHIC = 0, GRAFT_PERSPECTIVE = 1 }; /* mainTracing.c:23-24 */
enum { GRAFT_ANTIALIASING_OFF = 0, GRAFT_ANTIALIASING_ON = 1 }; /* mainTracing.c:26-27 */

/* pixel formats of the frame a trace writes */
enum {
    GRAFT_OUT_F64_RGB = 0,  /* 3 doubles per pixel: exactly the values the reference passes to pixSetRGB */
    GRAFT_OUT_F32_RGBA = 1, /* float4 per pixel (r, g, b, 1): one 128-bit store per pixel */
    GRAFT_OUT_U8_RGBA = 2   /* 4 bytes per pixel, q(v) = (uint8)(min(max(v,0),1)*255 + 0.5), alpha 255 */
};

/* One shape: the reference's `shape` (shapes.c:14-24) with the function pointers replaced by a kind tag and
 * the heap `unif` array inlined.  sphere: geom = {cx, cy, cz, radius, -, -} (shapes.c:144-145);
 * plane: geom = {px, py, pz, nx, ny, nz} (shapes.c:114-115). */
typedef struct GraftShape {
    int32_t kind;
    int32_t reserved;
    double geom[6];
    double color[3];     /* unif+4 (sphere, shapes.c:147) or unif+6 (plane, shapes.c:116) */
    double reflectivity; /* shapes.c:20 */
    double ambientLight; /* shapes.c:21 */
    double refrIndex;    /* shapes.c:22 */
    double transparency; /* shapes.c:23; > 0 switches on refraction()/fresnel() (mainTracing.c:232-294) */
} GraftShape;

/* One light: the reference's `light` (light.c:1-6); the `shape **avoid` list becomes a slice
 * [avoidOffset, avoidOffset + avoidNum) of GraftScene.avoidIndex holding indices into GraftScene.shapes. */
typedef struct GraftLight {
    double color[3];
    double loc[3];
    int32_t avoidOffset;
    int32_t avoidNum;
} GraftLight;

/* numShapes/shapes[], numLights/lights[], backgroundColor, maxDepth: mainTracing.c:35-39, :63-64 */
typedef struct GraftScene {
    int32_t numShapes;
    int32_t numLights;
    const GraftShape *shapes;
    const GraftLight *lights;
    const int32_t *avoidIndex;
    int32_t avoidIndexLen;
    int32_t maxDepth;
    double backgroundColor[3];
} GraftScene;

/* Camera globals as they stand after camInitialize (mainTracing.c:54-62, :375-384) plus the loop headers of
 * launchRays: colOffsets/rowOffsets are the values `i` and `j` take at mainTracing.c:392-393 (accumulated
 * doubles: fill them with graftLoopOffsets), subStepX/Y the sub-pixel steps of mainTracing.c:405-406. */
typedef struct GraftView {
    double screenCenter[3];
    double camPos[3];
    double viewDir[3];
    double lrVec[3];
    double udVec[3];
    int32_t projectionType;
    int32_t aliasing;
    int32_t numPasses;
    int32_t reserved;
    double subStepX; /* screenWidth / WIDTH / numPasses  */
    double subStepY; /* screenHeight / WIDTH / numPasses (the reference divides by WIDTH here too) */
    int32_t nCols;   /* trip count of the outer loop, mainTracing.c:392 */
    int32_t nRows;   /* trip count of the inner loop, mainTracing.c:393 */
    const double *colOffsets;
    const double *rowOffsets;
} GraftView;

/* Ray statistics of a trace.  `rays`, `tests`: what the REFERENCE does for these pixels -- closest-hit queries (one
 * per rayIntersectAvoid call, mainTracing.c:128: primary + shadow + reflection) and ray-shape evaluations (one per
 * shapes[k]->intersection call, :139).  `raysTraced`: the closest-hit queries the library made.  It makes fewer:
 * lighting() (:196-229) lets every unshadowed light overwrite the one before, so the library asks the lights last
 * to first and stops at the first unshadowed one; the shadow rays of the lights before it are never shot. */
typedef struct GraftCounters {
    uint64_t rays;
    uint64_t tests;
    uint64_t raysTraced;
    uint64_t reserved;
} GraftCounters;

typedef struct GraftJob GraftJob; /* a scene + view packed and resident in device memory */

/* ---- library state ---------------------------------------------------------------------------------- */

/* Number of HIP devices visible, or a negative GRAFT_ERR code.  (No reference counterpart: the reference
 * has no device.) */
GRAFT_API int graftDeviceCount(void);

/* Bind this process to one GPU and create the library's stream.  Idempotent for the same device.
 * Stands where pixInitialize stands in the reference's start-up (mainTracing.c:638): call once before
 * the first launchRays. */
GRAFT_API int graftInit(int device);

/* Release what graftInit and graftRenderRows hold (stream, reusable device buffers).  Destroy jobs first: a job
 * still alive keeps its device memory.  Stands where objectDestroy stands, mainTracing.c:646-648. */
GRAFT_API void graftShutdown(void);

/* Text of the last error; never NULL. */
GRAFT_API const char *graftLastError(void);

/* Diagnostic settings: for A/B timing and for the tests that run every kernel variant.  None of them changes a
 * result.  They start from the environment, which is read ONCE, at the first graftInit or graftDiagnostic* call
 * (GRAFT_SCREEN, GRAFT_GROUP, GRAFT_PER_TICKET, GRAFT_RESERVED_CUS); afterwards only graftDiagnosticSet changes them, and
 * graftDiagnosticGet reports the values in force.  A setting applies to jobs created (SCREEN_CAP, GROUP_SIZE) or
 * traces queued (TILES_PER_TICKET) after the call.  (No reference counterpart.) */
enum {
    GRAFT_DIAG_SCREEN_CAP = 0,       /* 0..7: highest screen level a job may use (graftJobScreenLevel); default 7 */
    GRAFT_DIAG_GROUP_SIZE = 1,       /* spheres per group of the two-level screens: 0 = chosen per scene, else 4..64, a multiple of 4 */
    GRAFT_DIAG_TILES_PER_TICKET = 2, /* 8x8 tiles a wave draws from the ticket counters at once: 0 = chosen per launch */
    GRAFT_DIAG_TABLE_BUILDS = 4,     /* read-only: screen tables built so far (shows what a re-render re-used) */
    GRAFT_DIAG_RESERVED_CUS = 5,     /* CUs a trace's persistent grid leaves free, 0..128; default 0 (graftRenderFrameMulti uses at least 4) */
    GRAFT_DIAG_COPY_PIECE_KB = 6,    /* graftRenderRows traces and copies a large frame in pieces of about this many KiB (at most 16 pieces); 0 = one piece; default -1 = chosen per frame */
    GRAFT_DIAG_EMULATED_DEVICES = 7  /* tests on a machine with too few GPUs: with n > 0 graftInitDevices may bind up to n devices, device d
                                      * standing for GPU d mod (GPUs present) with streams, buffers and a job of its own; pieces then reach the
                                      * root by device-to-device copies instead of RCCL; default 0 */
};
GRAFT_API int graftDiagnosticSet(int which, int value);
GRAFT_API int graftDiagnosticGet(int which); /* -1 for an unknown setting */

/* ---- host-side steps of launchRays (pure CPU, bit-for-bit the reference's arithmetic) --------------- */

/* The values of `for (double i = -extent/2; i < extent/2; i = i + extent/n)` (mainTracing.c:392-393).
 * Writes at most cap values, returns the trip count (may be n+1 after zooming: the counter is an
 * accumulated double) or -1 if the loop would not terminate (extent <= 0 or not finite). */
GRAFT_API int graftLoopOffsets(double extent, int n, double *out, int cap);

/* getScreenCoordX/Y (mainTracing.c:89-100): pixel coordinate of world offset `coord`, n = WIDTH|HEIGHT. */
GRAFT_API int graftScreenCoord(double coord, int n, double extent);

/* camInitialize (mainTracing.c:375-384) via getNewPoint (:104-114). */
GRAFT_API void graftCamInitialize(const double target[3], const double viewDir[3], double targetToScreen,
                                  double screenToCam, double screenCenter[3], double camPos[3]);

/* ---- page-locked host memory --------------------------------------------------------------------------- */

/* A host buffer the GPU can copy into directly (hipHostMalloc): a frame buffer allocated here makes the copy back
 * in graftRenderRows one DMA transfer instead of a staged one.  The launchRays() drop-in keeps its frame in one.
 * (No reference counterpart: the reference's frame lives inside the display library, 000pixel.h:53-55.) */
GRAFT_API int graftHostAlloc(size_t bytes, void **out);
GRAFT_API void graftHostFree(void *p);

/* ---- the path -------------------------------------------------------------------------------------- */

/* Validate, pack (shapes to per-kind geometry arrays + materials, avoid lists to indices) and upload a scene
 * and view.  The job owns device memory only; no host pointer is retained.  Replaces the implicit "read the
 * globals" of launchRays (mainTracing.c:387-391). */
GRAFT_API int graftJobCreate(const GraftScene *scene, const GraftView *view, GraftJob **job);

GRAFT_API void graftJobDestroy(GraftJob *job);

/* Which screen the job's kernels use in front of the exact sphere test (diagnostics; no reference counterpart):
 * 0 = none (exact double loop), 1 = single-precision screen per sphere, 2 = two-level ball screen, 3 = two-level
 * screens per ray family, 4 = three levels with cone member sections in global memory, 5 = level 4 with a uniform
 * grid for the rays behind a mirror, 6 = candidate lists (per tile for primary rays, per sphere and light for shadow
 * rays) in LDS, 7 = candidate lists in global memory plus the grid for the rays behind a mirror.  -1 for a null job. */
GRAFT_API int graftJobScreenLevel(const GraftJob *job);

/* Trace loop rows [row0, row0+nRows) of the job's frame into DEVICE memory: `frame` holds nRows*nCols pixels
 * of `format`, row-major in iy then ix; `hitIndex` (optional, may be NULL) nRows*nCols int32 closest-shape
 * indices (-1 = miss) of each pixel's first primary ray (the mask of rayIntersect, mainTracing.c:157, over
 * the rays of :425-437 when anti-aliasing is off); `counters` (optional) one GraftCounters in device memory,
 * ADDED to.  Asynchronous on `stream` (a hipStream_t; NULL = the library's stream).  Traces of one job may be
 * queued back to back on one stream; on different streams they must not overlap in time (a job has one set of tile
 * counters).  At most 0x7fff0000 tiles of 8x8 pixels per call.  This is the double loop of mainTracing.c:392-441
 * for those rows.  (The frame is all that launchRays produces; a call that passes neither `hitIndex` nor `counters`
 * may run a leaner kernel than one that does -- the frame is the same bit for bit.) */
GRAFT_API int graftJobTrace(GraftJob *job, int row0, int nRows, int format, void *frame, int32_t *hitIndex,
                            GraftCounters *counters, void *stream);

/* Block until everything queued on the library's stream has finished. */
GRAFT_API int graftSynchronize(void);

/* One-shot convenience for host callers: pack and upload scene and view, trace rows [row0,row0+nRows) and copy the
 * result to HOST buffers (frame in `format`; hitIndex and counters optional).  This is what a launchRays() drop-in
 * calls once per frame (mainTracing.c:630, :642).  State kept between calls, until graftShutdown: the device-side
 * frame buffers (grow-only), and the packed job -- a call whose scene, view and settings equal the previous call's
 * byte for byte packs and uploads nothing; one that differs re-packs into the same device block, re-using the screen
 * tables whose inputs did not change (a key press that toggles anti-aliasing or depth rebuilds none, one that turns
 * the view rebuilds the camera's only).  A frame of more than about 100 MiB is traced in up to 16 pieces of rows, each
 * copied to the host (on a stream of its own) while the next ones are traced; give it a page-locked frameHost
 * (graftHostAlloc), or the copies cannot overlap anything. */
GRAFT_API int graftRenderRows(const GraftScene *scene, const GraftView *view, int row0, int nRows, int format,
                              void *frameHost, int32_t *hitIndexHost, GraftCounters *countersHost);

/* graftRenderRows for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc): no
 * counterpart in the reference.  (Not for use under HIP graph capture: the call waits on an event and stages the
 * upload through one page-locked block, so a captured call could neither be repeated nor replayed.)  It calls no hipMalloc, hipFree, hipHostMalloc or other
 * allocating HIP routine: the device block a trace reads (tables + ticket counters) lives in a static __device__
 * array of GRAFT_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends, reached
 * through the address graftInit looked up for that GPU, and the upload goes through a page-locked block graftInit
 * allocated.  (Packing the scene on the host uses the C++ heap, as graftJobCreate does.)  Inputs are host structs as
 * for graftRenderRows; outputs are DEVICE buffers of the caller: `frameDevice` nRows*nCols pixels of `format`,
 * `hitIndexDevice` and `countersDevice` optional as for graftJobTrace (counters are ADDED to).  Asynchronous on
 * `stream` (NULL = the library's stream).  A scene + view that packs to more than the scratch block is refused with
 * GRAFT_ERR_UNSUPPORTED (4096 spheres with 4 lights at 16384 x 16384 pack to 1.2 MB).  Every call on a GPU uses that
 * GPU's one block, which keeps the previous call's contents until the next upload: calls on one stream follow each
 * other safely; calls of this entry point that overlap in time on one GPU (different streams) race. */
#define GRAFT_NOALLOC_SCRATCH_BYTES (4u << 20)
GRAFT_API int graftRenderRowsNoAlloc(const GraftScene *scene, const GraftView *view, int row0, int nRows, int format,
                                     void *frameDevice, int32_t *hitIndexDevice, GraftCounters *countersDevice,
                                     void *stream);

/* ---- several GPUs driven by one process ------------------------------------------------------------------ */

/* The reference is single-threaded and knows no device (mainTracing.c:387-443); these entry points let its one host
 * thread spread a frame over the GPUs of a node.  Rows are cut into tile-aligned pieces dealt round-robin (rows differ
 * in cost; dealing pieces of every part of the frame to every GPU evens that out), every GPU holds the whole packed
 * scene, and the pieces are assembled in GPU 0's memory by RCCL send/recv over xGMI (one communicator per GPU from
 * ncclCommInitAll; librccl.so is loaded when more than one GPU is asked for), each round of transfers hidden behind
 * the tracing of the next round's pieces.
 *
 * graftInitDevices binds GPUs 0 .. nDevices-1 (GPU 0 is the root and the device of the single-GPU entry points; it is
 * graftInit(0) when nDevices is 1); if it fails, no multi-device frame can be started until a later call succeeds; a later
 * call with fewer devices goes on with the first nDevices of those bound.
 * graftFramePlan is the piece list (pure host arithmetic; returns the number of pieces, writes at most `cap`
 * entries per array, any array may be NULL): piece k covers rows [firstRow[k], firstRow[k] + rows[k]), belongs to
 * GPU owner[k] and starts at row rowInOwner[k] of that GPU's own buffer of ownedRows[owner[k]] rows (ownedRows has
 * nDevices entries); graftFramePieces is the same without the last two.  graftRenderFrameMulti is graftRenderRows for
 * the whole frame over all bound GPUs, host buffers out; with one GPU bound it IS graftRenderRows.  The scene is packed
 * once and uploaded to every GPU; traces leave 4 CUs of every GPU to the transfer kernels; the rows of every finished
 * round leave the root for the host (page-locked memory: graftHostAlloc) while the later rounds are traced.  graftMultiStats reports,
 * for one GPU of the last multi-GPU frame, the time from its first kernel to its last (traceMs) and from its last
 * kernel to its last transfer (tailMs: the part of the exchange tracing did not hide; negative if none). */
GRAFT_API int graftInitDevices(int nDevices);
GRAFT_API int graftFramePlan(int nRows, int nDevices, int piecesPerDevice, int *firstRow, int *rows, int *owner,
                             int *rowInOwner, int *ownedRows, int cap);
GRAFT_API int graftFramePieces(int nRows, int nDevices, int piecesPerDevice, int *firstRow, int *rows, int *owner, int cap);
GRAFT_API int graftMultiStats(int device, float *traceMs, float *tailMs);
GRAFT_API int graftRenderFrameMulti(const GraftScene *scene, const GraftView *view, int format, int piecesPerDevice,
                                    void *frameHost, int32_t *hitIndexHost, GraftCounters *countersHost);

/* Write a host frame (row iy = 0 is the BOTTOM row, as pixSetRGB's y: 000pixel.h:53-55) as a binary PPM (P6, top
 * row first), quantising with q(v) = (uint8)(min(max(v,0),1)*255 + 0.5) where the format is not already 8 bit.
 * For headless runs: the reference can only show its frame in a window (000pixel.h:31-39). */
GRAFT_API int graftWritePPM(const char *path, const void *frameHost, int format, int nCols, int nRows);
/* The same frame as a PNG (8-bit RGB, uncompressed deflate blocks: no library needed); frames whose scanlines exceed
 * 2 GiB are refused with GRAFT_ERR_UNSUPPORTED. */
GRAFT_API int graftWritePNG(const char *path, const void *frameHost, int format, int nCols, int nRows);

/* Bytes per pixel of a GRAFT_OUT_* format, 0 if unknown. */
GRAFT_API size_t graftPixelBytes(int format);

/* Probe kernels used by the numerics tests: out[i] = sqrt(a[i]) and out[i] = a[i] / b[i] evaluated on the
 * device (host pointers, n elements).  The path relies on both being correctly rounded, as libm sqrt and
 * the x86 divide are for the reference (vector.c:67-82, shapes.c:58, :88). */
GRAFT_API int graftProbeSqrt(const double *a, double *out, size_t n);
GRAFT_API int graftProbeDiv(const double *a, const double *b, double *out, size_t n);
/* out[3i..3i+2] = vecUnit of v[3i..3i+2] (vector.c:74-82; left at zero when the length is not positive) as the hot
 * stages of the kernels evaluate it: the square root without the rescaling a vector of ordinary length never needs,
 * the three quotients through one refined reciprocal.  Must equal sqrt and `/` bit for bit. */
GRAFT_API int graftProbeUnit(const double *v, double *out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* GRAFTHIP_H */
