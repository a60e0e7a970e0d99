/*
 * launch_rays_dropin.c -- `void launchRays(void)` for the reference program, running on the GPU.
 *
 * Link this object (and libgrafthip.so) with mainTracing.c in place of the reference's own launchRays
 * (mainTracing.c:387-443).  It reads the same globals, leaves them as the reference would (camInitialize
 * updates screenCenter and camPos, :388), and makes the same pixSetRGB calls in the same order.
 * WIDTH and HEIGHT are compile-time constants in the reference (mainTracing.c:20-21): build this file with
 * the same values (-DGRAFT_WIDTH=..., -DGRAFT_HEIGHT=...).
 */
#include <stdio.h>
#include <stdlib.h>

#include "grafthip.h"
#include "rt_reference_types.h"

#ifndef GRAFT_WIDTH
#define GRAFT_WIDTH 512
#endif
#ifndef GRAFT_HEIGHT
#define GRAFT_HEIGHT 512
#endif
/* GPUs to spread a frame over (one host thread drives them all): 1 unless the build says otherwise; never more than
 * the machine has.  A 512 x 512 frame is far too small to gain from a second GPU; a 16384 x 16384 one is not. */
#ifndef GRAFT_DEVICES
#define GRAFT_DEVICES 1
#endif

/* The frame the GPU copies into: page-locked, kept from frame to frame (an interactive host traces one after every
 * key press, mainTracing.c:629-630), grown when the loop runs more iterations (513 per axis after zooming). */
static double *frameBuffer = NULL;
static size_t frameBufferBytes = 0;

static double *frame_buffer(size_t bytes) {
    if (bytes > frameBufferBytes) {
        void *p = NULL;
        graftHostFree(frameBuffer);
        frameBuffer = NULL;
        frameBufferBytes = 0;
        if (graftHostAlloc(bytes, &p) != GRAFT_OK) return NULL;
        frameBuffer = p;
        frameBufferBytes = bytes;
    }
    return frameBuffer;
}

static int fill_offsets(double extent, int n, double **out) {
    int cap = 2 * n + 32;
    *out = malloc(sizeof(double) * cap);
    return *out ? graftLoopOffsets(extent, n, *out, cap) : -1;
}

void launchRays(void) {
    GraftShape *gs = NULL;
    GraftLight *gl = NULL;
    int32_t *avoid = NULL;
    double *cols = NULL, *rows = NULL, *rgb = NULL;
    int nAvoid = 0, rc = GRAFT_OK;

    {
        int devices = graftDeviceCount();
        if (devices > GRAFT_DEVICES) devices = GRAFT_DEVICES;
        if ((rc = graftInitDevices(devices < 1 ? 1 : devices)) != GRAFT_OK) goto done;
    }
    graftCamInitialize(target, viewDir, targetToScreen, screenToCam, screenCenter, camPos); /* :388 */

    gs = calloc(numShapes > 0 ? numShapes : 1, sizeof *gs);
    gl = calloc(numLights > 0 ? numLights : 1, sizeof *gl);
    for (int i = 0; i < numLights; ++i) nAvoid += lights[i]->avoidNum;
    avoid = calloc(nAvoid > 0 ? nAvoid : 1, sizeof *avoid);
    if (!gs || !gl || !avoid) { rc = -1; goto done; }

    for (int k = 0; k < numShapes; ++k) {
        const shape *s = shapes[k];
        const int plane = s->intersection == planeIntersect;
        gs[k].kind = plane ? GRAFT_PLANE : GRAFT_SPHERE;
        for (int c = 0; c < (plane ? 6 : 4); ++c) gs[k].geom[c] = s->unif[c];
        for (int c = 0; c < 3; ++c) gs[k].color[c] = s->unif[(plane ? 6 : 4) + c];
        gs[k].reflectivity = s->reflectivity;
        gs[k].ambientLight = s->ambientLight;
        gs[k].refrIndex = s->refrIndex;
        gs[k].transparency = s->transparency;
    }
    nAvoid = 0;
    for (int i = 0; i < numLights; ++i) {
        const light *l = lights[i];
        for (int c = 0; c < 3; ++c) { gl[i].color[c] = l->color[c]; gl[i].loc[c] = l->loc[c]; }
        gl[i].avoidOffset = nAvoid;
        for (int a = 0; a < l->avoidNum; ++a) { /* pointer identity, as contains() (:116-123) */
            int idx = -1;
            for (int k = 0; k < numShapes; ++k)
                if (shapes[k] == l->avoid[a]) { idx = k; break; }
            avoid[nAvoid++] = idx;
        }
        gl[i].avoidNum = nAvoid - gl[i].avoidOffset;
    }

    GraftScene scene = {0};
    scene.numShapes = numShapes; scene.numLights = numLights;
    scene.shapes = gs; scene.lights = gl; scene.avoidIndex = avoid; scene.avoidIndexLen = nAvoid;
    scene.maxDepth = maxDepth;
    for (int c = 0; c < 3; ++c) scene.backgroundColor[c] = backgroundColor[c];

    GraftView view = {0};
    for (int c = 0; c < 3; ++c) {
        view.screenCenter[c] = screenCenter[c]; view.camPos[c] = camPos[c]; view.viewDir[c] = viewDir[c];
        view.lrVec[c] = lrVec[c]; view.udVec[c] = udVec[c];
    }
    view.projectionType = projectionType;
    view.aliasing = aliasing == GRAFT_ANTIALIASING_ON;
    view.numPasses = numPasses;
    if (view.aliasing) {
        view.subStepX = screenWidth / GRAFT_WIDTH / numPasses;  /* :405 */
        view.subStepY = screenHeight / GRAFT_WIDTH / numPasses; /* :406, WIDTH as in the reference */
    }
    view.nCols = fill_offsets(screenWidth, GRAFT_WIDTH, &cols);
    view.nRows = fill_offsets(screenHeight, GRAFT_HEIGHT, &rows);
    if (view.nCols <= 0 || view.nRows <= 0) { rc = -2; goto done; }
    view.colOffsets = cols; view.rowOffsets = rows;

    rgb = frame_buffer(sizeof(double) * 3 * (size_t)view.nCols * view.nRows);
    if (!rgb) { rc = GRAFT_ERR_HIP; goto done; }
    rc = graftRenderFrameMulti(&scene, &view, GRAFT_OUT_F64_RGB, 8, rgb, NULL, NULL); /* one GPU: graftRenderRows */
    if (rc != GRAFT_OK) goto done;

    /* the display stays on the host: same calls, same order as :392-393, :422, :439 */
    for (int a = 0; a < view.nCols; ++a) {
        const int x = graftScreenCoord(cols[a], GRAFT_WIDTH, screenWidth);
        for (int b = 0; b < view.nRows; ++b) {
            const double *p = rgb + 3 * ((size_t)b * view.nCols + a);
            pixSetRGB(x, graftScreenCoord(rows[b], GRAFT_HEIGHT, screenHeight), p[0], p[1], p[2]);
        }
    }
done:
    if (rc != GRAFT_OK) /* the reference has no error channel (void, nothing checked): say so and draw nothing */
        fprintf(stderr, "launchRays (GPU): error %d: %s\n", rc, rc > 0 ? graftLastError() : "host allocation or loop bounds");
    free(gs); free(gl); free(avoid); free(cols); free(rows);
}
