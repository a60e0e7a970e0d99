/*
 * Reference harness body  --  TEST INFRASTRUCTURE, never part of the product.
 *
 * This text is appended (by oracle/Makefile) AFTER the reference's own
 * mainTracing.c inside one translation unit, so that every function and global
 * of the reference (launchRays, rayIntersect, shapes[], lights[], viewDir ...)
 * is visible here.  Nothing of the reference is restated: the harness only
 *   - supplies the output sink of the path, i.e. the six pix* functions that
 *     mainTracing.c calls (000pixel.h:31-81).  The display library behind them
 *     is a macOS object file and is out of scope; pixSetRGB is where the path
 *     hands its result over, so the sink records what it is given;
 *   - exports small setters/getters around the reference's globals so a test
 *     can build scenes with the reference's own sphereInit/planeInit/lightInit
 *     and press the reference's own keys (handleKeyDown).
 *
 * Two builds exist (see oracle/Makefile):
 *   libref_default.so   mainTracing.c unmodified (512x512, 11 shapes, 2 lights)
 *   librefp_<W>x<H>.so  four constants patched on the fly (WIDTH, HEIGHT,
 *                       shapes[11], lights[2]) and mainTracing.c:323 made a
 *                       no-op, to run larger scenes and other frame sizes.
 */
#include <stdint.h>
#include <string.h>

#ifndef REF_MAXSHAPES
#define REF_MAXSHAPES 11
#endif
#ifndef REF_MAXLIGHTS
#define REF_MAXLIGHTS 2
#endif

static double *sinkFrame = NULL;      /* [HEIGHT][WIDTH][3], row y, column x */
static int32_t *sinkLog = NULL;       /* optional (x,y) pairs in call order   */
static long sinkLogCap = 0, sinkCalls = 0, sinkOutOfRange = 0;
static int shapesLive = 0, lightsLive = 0;
static shape **avoidLists[REF_MAXLIGHTS];

/* ---- output sink: the pix* functions used by mainTracing.c:638-645, :422, :439, :629 ---- */
int pixInitialize(int width, int height, char *name) { (void)width; (void)height; (void)name; return 0; }
void pixRun(void) {}
void pixSetKeyDownHandler(void (*handler)(int, int, int, int, int)) { (void)handler; }
void pixSetKeyRepeatHandler(void (*handler)(int, int, int, int, int)) { (void)handler; }
void pixClearRGB(double red, double green, double blue) {
    if (!sinkFrame) return;
    for (long p = 0; p < (long)WIDTH * HEIGHT; ++p) {
        sinkFrame[3 * p] = red; sinkFrame[3 * p + 1] = green; sinkFrame[3 * p + 2] = blue;
    }
}
void pixSetRGB(int x, int y, double red, double green, double blue) {
    if (sinkLog && sinkCalls < sinkLogCap) { sinkLog[2 * sinkCalls] = x; sinkLog[2 * sinkCalls + 1] = y; }
    sinkCalls++;
    if (x < 0 || x >= WIDTH || y < 0 || y >= HEIGHT) { sinkOutOfRange++; return; }
    if (!sinkFrame) return;
    double *px = sinkFrame + 3 * ((long)y * WIDTH + x);
    px[0] = red; px[1] = green; px[2] = blue;
}

/* ---- exported helpers ---- */
int ref_width(void) { return WIDTH; }
int ref_height(void) { return HEIGHT; }
int ref_max_shapes(void) { return REF_MAXSHAPES; }
int ref_max_lights(void) { return REF_MAXLIGHTS; }
long ref_sink_calls(void) { return sinkCalls; }
long ref_sink_out_of_range(void) { return sinkOutOfRange; }

void ref_clear_scene(void) {
    for (int k = 0; k < shapesLive; ++k) objectDestroy(shapes[k]);
    for (int k = 0; k < lightsLive; ++k) { free(lights[k]); free(avoidLists[k]); avoidLists[k] = NULL; }
    shapesLive = lightsLive = 0;
    numShapes = 0; numLights = 0;
}

/* the reference's own start-up sequence, mainTracing.c:634-641 */
void ref_default_scene(void) {
    ref_clear_scene();
    screenWidth = 512; screenHeight = 512;
    backgroundColor[0] = backgroundColor[1] = backgroundColor[2] = 0;
    projectionType = PERSPECTIVE; aliasing = ANTIALIASING_ON; maxDepth = maxDepthLow; numPasses = 6;
    initializeShapes();
    initializeLights();
    shapesLive = 11; lightsLive = 2;
    numShapes = 11; numLights = 1;
}

void ref_set_flags(int projection, int aa, int passes, int depth) {
    projectionType = projection; aliasing = aa; numPasses = passes; maxDepth = depth;
}
void ref_set_counts(int nShapes, int nLights) { numShapes = nShapes; numLights = nLights; }
void ref_set_screen(double w, double h) { screenWidth = w; screenHeight = h; }
void ref_set_background(const double rgb[3]) { for (int c = 0; c < 3; ++c) backgroundColor[c] = rgb[c]; }

int ref_add_sphere(double center[3], double radius, double reflectivity, double rgb[3], double ambient,
                   double refrIndex, double transparency) {
    if (shapesLive >= REF_MAXSHAPES) return -1;
    sphereSetup(radius, center, shapesLive, rgb, reflectivity, ambient, refrIndex, transparency);
    numShapes = ++shapesLive;
    return shapesLive - 1;
}
int ref_add_plane(double point[3], double normal[3], double reflectivity, double rgb[3], double ambient,
                  double refrIndex, double transparency) {
    if (shapesLive >= REF_MAXSHAPES) return -1;
    planeSetup(normal, point, shapesLive, rgb, reflectivity, ambient, refrIndex, transparency);
    numShapes = ++shapesLive;
    return shapesLive - 1;
}
/* avoid list given as indices into shapes[]; stored as the shape** the reference expects (light.c:8-11) */
int ref_add_light(double color[3], double loc[3], const int32_t *avoidIdx, int nAvoid) {
    if (lightsLive >= REF_MAXLIGHTS) return -1;
    light *l = malloc(sizeof(light));
    lightInit(l, color, loc);
    if (nAvoid > 0) {
        shape **lst = malloc(sizeof(shape *) * nAvoid);
        for (int a = 0; a < nAvoid; ++a) lst[a] = shapes[avoidIdx[a]];
        avoidLists[lightsLive] = lst;
        lightSetAvoid(l, lst, nAvoid);
    }
    lights[lightsLive] = l;
    numLights = ++lightsLive;
    return lightsLive - 1;
}
/* overwrite the four material scalars of shapes[k] (shapes.c:20-23) */
void ref_set_material(int k, double reflectivity, double ambient, double refrIndex, double transparency) {
    shapes[k]->reflectivity = reflectivity; shapes[k]->ambientLight = ambient;
    shapes[k]->refrIndex = refrIndex; shapes[k]->transparency = transparency;
}
void ref_scene_initialize(double tgt[3], double toScreen, double toCam) { sceneInitialize(tgt, toScreen, toCam); }
void ref_rotate_view(double theta, double axis[3]) { rotateView(theta, axis); }

/* out[27]: target3 viewDir3 lrVec3 udVec3 screenCenter3 camPos3 | targetToScreen screenToCam screenWidth
 * screenHeight | projectionType aliasing numPasses maxDepth numShapes (as doubles) */
void ref_get_state(double out[27]) {
    memcpy(out, target, 24); memcpy(out + 3, viewDir, 24); memcpy(out + 6, lrVec, 24);
    memcpy(out + 9, udVec, 24); memcpy(out + 12, screenCenter, 24); memcpy(out + 15, camPos, 24);
    out[18] = targetToScreen; out[19] = screenToCam; out[20] = screenWidth; out[21] = screenHeight;
    out[22] = projectionType; out[23] = aliasing; out[24] = numPasses; out[25] = maxDepth; out[26] = numShapes;
}

static void sinkBegin(double *frame, int32_t *log, long logCap) {
    sinkFrame = frame; sinkLog = log; sinkLogCap = logCap; sinkCalls = 0; sinkOutOfRange = 0;
}

/* one frame through the reference's launchRays (mainTracing.c:387); frame is caller-initialised */
void ref_launch(double *frame, int32_t *log, long logCap) {
    sinkBegin(frame, log, logCap);
    launchRays();
    sinkFrame = NULL; sinkLog = NULL;
}

/* one key press through the reference's handleKeyDown (mainTracing.c:552): mutates the globals, clears the
 * frame to the background colour and renders */
void ref_key(int key, int shift, double *frame, int32_t *log, long logCap) {
    sinkBegin(frame, log, logCap);
    handleKeyDown(key, shift, 0, 0, 0);
    sinkFrame = NULL; sinkLog = NULL;
}

/* closest-shape index of every anti-aliasing-off primary ray: rays are produced by the reference's own
 * helpers in the order of mainTracing.c:425-437 and handed to its rayIntersect (:157).  idx is
 * [HEIGHT][WIDTH], cells never addressed keep the caller's fill value. */
void ref_primary_index(int32_t *idx) {
    camInitialize(target, targetToScreen, screenToCam);
    for (double i = -screenWidth / 2; i < screenWidth / 2; i = i + screenWidth / WIDTH) {
        for (double j = -screenHeight / 2; j < screenHeight / 2; j = j + screenHeight / HEIGHT) {
            double colPoint[3], origin[3], dir[3], hitLoc[3], hitNormal[3];
            getNewPoint(screenCenter, lrVec, i, colPoint);
            getNewPoint(colPoint, udVec, j, origin);
            if (projectionType == ORTHOGRAPHIC) {
                double back[3] = {-viewDir[0], -viewDir[1], -viewDir[2]};
                vecUnit(3, back, dir);
            } else {
                vecSubtract(3, origin, camPos, dir);
                vecUnit(3, dir, dir);
            }
            int x = getScreenCoordX(i), y = getScreenCoordY(j);
            int hit = rayIntersect(origin, dir, hitLoc, hitNormal);
            if (x >= 0 && x < WIDTH && y >= 0 && y < HEIGHT) idx[(long)y * WIDTH + x] = hit;
        }
    }
}
