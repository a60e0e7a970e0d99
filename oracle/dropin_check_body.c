/*
 * Drop-in check  --  TEST INFRASTRUCTURE.  Built by oracle/Makefile into oracle/_ref/dropin_check.
 *
 * This text follows the reference's mainTracing.c (whose own launchRays has been renamed
 * launchRays_reference_cpu by a macro) and oracle/ref_harness_body.c in one translation unit, and is linked
 * with the product's launchRays() drop-in (raytracing_amd/csrc/launch_rays_dropin.c) and libgrafthip.so.
 * For a list of program states reached with the reference's OWN functions (initializeShapes,
 * initializeLights, handleKeyDown ...) it traces one frame with the reference's CPU launchRays and one with
 * the GPU drop-in -- both read the same globals and both deliver through the same pixSetRGB sink -- and
 * compares every byte of the two frames and the two (x, y) call sequences.  Exit status 0 = all identical.
 */
void launchRays(void); /* the GPU drop-in */

static int compareFrames(const char *what) {
    static double cpuFrame[HEIGHT][WIDTH][3], gpuFrame[HEIGHT][WIDTH][3];
    static int32_t cpuLog[(WIDTH + 2) * (HEIGHT + 2) * 2], gpuLog[(WIDTH + 2) * (HEIGHT + 2) * 2];
    const long cap = (long)(WIDTH + 2) * (HEIGHT + 2);
    for (long p = 0; p < (long)WIDTH * HEIGHT * 3; ++p) { (&cpuFrame[0][0][0])[p] = -1; (&gpuFrame[0][0][0])[p] = -1; }
    sinkBegin(&cpuFrame[0][0][0], cpuLog, cap);
    launchRays_reference_cpu();
    long cpuCalls = sinkCalls, cpuOut = sinkOutOfRange;
    sinkBegin(&gpuFrame[0][0][0], gpuLog, cap);
    launchRays();
    long gpuCalls = sinkCalls, gpuOut = sinkOutOfRange;
    sinkFrame = NULL; sinkLog = NULL;
    int same = cpuCalls == gpuCalls && cpuOut == gpuOut &&
               memcmp(cpuFrame, gpuFrame, sizeof cpuFrame) == 0 &&
               memcmp(cpuLog, gpuLog, sizeof(int32_t) * 2 * (cpuCalls < cap ? cpuCalls : cap)) == 0;
    printf("%-44s calls %ld/%ld outside %ld/%ld  %s\n", what, cpuCalls, gpuCalls, cpuOut, gpuOut,
           same ? "IDENTICAL" : "DIFFERENT");
    return same ? 0 : 1;
}

/* state change of a key press: the reference's handler also traces a frame (on the CPU); that frame is dropped */
static void press(int key, int shift) {
    sinkBegin(NULL, NULL, 0);
    handleKeyDown(key, shift, 0, 0, 0);
}

int main(void) {
    int bad = 0;
    ref_default_scene();
    ref_set_flags(PERSPECTIVE, ANTIALIASING_OFF, 6, 4);
    bad += compareFrames("start-up scene, anti-aliasing off");
    ref_set_flags(PERSPECTIVE, ANTIALIASING_ON, 2, 4);
    bad += compareFrames("start-up scene, 2x2 supersampling");
    ref_set_flags(ORTHOGRAPHIC, ANTIALIASING_OFF, 6, 10);
    bad += compareFrames("orthographic, maxDepth 10");
    ref_set_flags(PERSPECTIVE, ANTIALIASING_OFF, 6, 4);
    ref_set_counts(11, 2);
    bad += compareFrames("both lights");
    ref_set_counts(1, 1);
    bad += compareFrames("numShapes = 1");
    ref_set_counts(11, 1);
    press(263, 0);
    bad += compareFrames("after LEFT (rotateView)");
    press(265, 0);
    bad += compareFrames("after UP (rotateView about lrVec)");
    press(265, 1);
    bad += compareFrames("after shift+UP (zoom out, 513 iterations)");
    press(264, 1);
    press(264, 1);
    bad += compareFrames("after 2x shift+DOWN (zoom in)");
    press(69, 0);
    bad += compareFrames("after E (maxDepth 10)");
    printf(bad ? "%d state(s) differ\n" : "all states identical\n", bad);
    return bad ? 1 : 0;
}
