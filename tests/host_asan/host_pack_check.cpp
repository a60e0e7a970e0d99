// host_pack_check.cpp -- the host side of libgrafthip.so (packJob and the tables it builds, the pure-host entry points)
// compiled WITHOUT HIP, with AddressSanitizer and UndefinedBehaviorSanitizer, and driven by well-formed random scenes
// at every screen cap and group size and by hostile inputs.  The reference checks nothing (malloc unchecked,
// shapes.c:105-106; counts trusted, mainTracing.c:538); the library's contract is an error code and no wild access.
// Besides what the sanitizers see, every packed job is checked structurally: offsets inside the image, list entries
// inside the scene, cells inside the grid.  Exit status 0 = everything held.
//     make -C tests/host_asan && tests/host_asan/host_pack_check [scenes]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <string>
#include <vector>

#include "host_job.h"
#include "host_steps.h"

using namespace graft;

static int failures = 0;
#define EXPECT(cond, ...)                                   \
    do {                                                    \
        if (!(cond)) {                                      \
            ++failures;                                     \
            fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                   \
            fprintf(stderr, "\n");                          \
        }                                                   \
    } while (0)

struct Rng {
    uint64_t state;
    uint64_t next() {
        state += 0x9E3779B97F4A7C15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform(double a, double b) { return a + (b - a) * ((double)(next() >> 11) * 0x1p-53); }
    int pick(int n) { return (int)(next() % (uint64_t)n); }
};

struct Scene {
    std::vector<GraftShape> shapes;
    std::vector<GraftLight> lights;
    std::vector<int32_t> avoid;
    std::vector<double> cols, rows;
    GraftScene scene;
    GraftView view;
    void bind() {
        scene.numShapes = (int32_t)shapes.size();
        scene.numLights = (int32_t)lights.size();
        scene.shapes = shapes.data();
        scene.lights = lights.data();
        scene.avoidIndex = avoid.data();
        scene.avoidIndexLen = (int32_t)avoid.size();
        view.nCols = (int32_t)cols.size();
        view.nRows = (int32_t)rows.size();
        view.colOffsets = cols.data();
        view.rowOffsets = rows.data();
    }
};

static GraftShape sphere(double x, double y, double z, double r, double refl = 0, double transp = 0) {
    GraftShape s;
    memset(&s, 0, sizeof s);
    s.kind = GRAFT_SPHERE;
    s.geom[0] = x; s.geom[1] = y; s.geom[2] = z; s.geom[3] = r;
    s.color[0] = .5; s.color[1] = .6; s.color[2] = .7;
    s.reflectivity = refl; s.ambientLight = 0.1; s.refrIndex = 1.3; s.transparency = transp;
    return s;
}

static void frame(Scene &S, int w, int h, double extent, int projection, int passes) {
    S.cols.assign((size_t)(w + 2), 0.0);
    S.rows.assign((size_t)(h + 2), 0.0);
    const int nc = loopOffsets(extent, w, S.cols.data(), w + 2), nr = loopOffsets(extent, h, S.rows.data(), h + 2);
    S.cols.resize((size_t)(nc > 0 ? nc : 1));
    S.rows.resize((size_t)(nr > 0 ? nr : 1));
    memset(&S.view, 0, sizeof S.view);
    const double target[3] = {0, 0, 0}, viewDir[3] = {0, 0, -1};
    camInitializeHost(target, viewDir, 256, 1000, S.view.screenCenter, S.view.camPos);
    S.view.viewDir[2] = -1; S.view.lrVec[0] = 1; S.view.udVec[1] = 1;
    S.view.projectionType = projection;
    S.view.aliasing = passes > 0;
    S.view.numPasses = passes > 0 ? passes : 1;
    S.view.subStepX = extent / w / S.view.numPasses;
    S.view.subStepY = extent / w / S.view.numPasses;
}

static Scene randomScene(uint64_t seed) {
    Rng rng{0xA5A50000ull + seed};
    Scene S;
    const int counts[] = {0, 1, 3, 31, 32, 64, 70, 130, 300, 700, 1500};
    const int n = counts[rng.pick(11)];
    const double box = rng.pick(2) ? 400.0 : 150.0;
    const double rmax = rng.pick(3) == 0 ? 120.0 : 12.0;
    const bool mirrors = rng.pick(2) != 0;
    for (int k = 0; k < n; ++k)
        S.shapes.push_back(sphere(rng.uniform(-box, box), rng.uniform(-box, box), 100 + rng.uniform(-box, box), rng.uniform(1.0, rmax),
                                  mirrors && rng.pick(2) ? 0.5 : 0.0));
    if (rng.pick(4) == 0) {  // two planes
        GraftShape p = sphere(0, -300, 0, 0);
        p.kind = GRAFT_PLANE;
        p.geom[3] = 0; p.geom[4] = 1; p.geom[5] = 0;
        S.shapes.push_back(p);
        S.shapes.push_back(p);
    }
    if (rng.pick(5) == 0 && !S.shapes.empty()) S.shapes[0].transparency = 0.7;
    const int nl = 1 + rng.pick(9);
    for (int i = 0; i < nl; ++i) {
        GraftLight l;
        memset(&l, 0, sizeof l);
        l.color[0] = l.color[1] = l.color[2] = 1;
        l.loc[0] = rng.uniform(-300, 300); l.loc[1] = rng.uniform(-300, 450); l.loc[2] = rng.uniform(-200, 300);
        if (rng.pick(2) && !S.shapes.empty()) {
            l.avoidOffset = (int32_t)S.avoid.size();
            l.avoidNum = rng.pick(4);
            for (int a = 0; a < l.avoidNum; ++a) S.avoid.push_back(rng.pick((int)S.shapes.size() + 2) - 1);  // (some out of range)
        }
        S.lights.push_back(l);
    }
    memset(&S.scene, 0, sizeof S.scene);
    S.scene.maxDepth = 1 + rng.pick(6);
    const int sizes[][2] = {{48, 40}, {8, 8}, {1, 1}, {129, 65}, {512, 16}};
    const int which = rng.pick(5);
    const double extents[] = {512.0, 512.0 * 0.9, 512.0 * 1.1, 3.0, 40000.0};
    frame(S, sizes[which][0], sizes[which][1], extents[rng.pick(5)], rng.pick(4) ? GRAFT_PERSPECTIVE : GRAFT_ORTHOGRAPHIC,
          rng.pick(3) ? 0 : 1 + rng.pick(9));
    S.bind();
    return S;
}

// everything a kernel would index with must lie inside what was packed
static void checkStructure(const PackedJob &J, const Scene &S, const char *what) {
    const TraceParams &P = J.P;
    const size_t total = J.image.size();
    EXPECT(J.oCursor + sizeof(int) * kCursors * kCursorStride <= total, "%s: ticket counters outside the image", what);
    EXPECT(J.oGeomF + sizeof(float) * (size_t)P.filterFloats <= J.oMember, "%s: screen table overlaps what follows", what);
    EXPECT(J.ldsBytes <= kLdsMax, "%s: %zu bytes of LDS", what, J.ldsBytes);
    EXPECT(J.block == kSmallBlock || J.block == 2 * kSmallBlock || J.block == kLargeBlock, "%s: block %d", what, J.block);
    const size_t n = (size_t)P.numSpheres;
    if (J.screen >= 6) {
        const unsigned char *img = J.image.data() + J.oGeomF;
        const size_t bytes = sizeof(float) * (size_t)P.filterFloats, nL = (size_t)P.numLights;
        const size_t staged = sizeof(float) * (size_t)P.ldsHeadFloats;
        EXPECT(staged <= bytes && (J.screen == 7 || staged == bytes), "%s: staged part", what);
        EXPECT((size_t)P.listTileOff == staged && staged + (size_t)(J.block / 64) * 2 * kTileListCap <= J.ldsBytes, "%s: tile lists", what);
        if (P.sampleTableOff >= 0)
            EXPECT((size_t)P.sampleTableOff + (size_t)(J.block / 64) * 128 * (size_t)P.numPasses <= J.ldsBytes && P.numPasses <= kSampleTablePasses,
                   "%s: sample tables", what);
        const size_t members = (size_t)P.listGroups * (size_t)P.groupSize;
        EXPECT((size_t)P.listGroupsOff + 16 * (size_t)P.listGroups <= bytes && (size_t)P.listSinesOff + 4 * (size_t)P.listGroups <= bytes &&
               (size_t)P.listMembersOff + 16 * members <= bytes && (size_t)P.listMemberSinesOff + 4 * members <= bytes, "%s: camera table", what);
        // what the tile-list scan reads from LDS at either level lies inside the staged part
        EXPECT((size_t)P.listSinesOff + 4 * (size_t)P.listGroups <= 4 * (size_t)P.ldsHeadFloats, "%s: camera groups staged", what);
        EXPECT((size_t)P.listShapeOff + 2 * (size_t)P.listGroups * P.groupSize <= bytes, "%s: member numbers", what);
        EXPECT((size_t)P.listIndexOff + 4 * n * nL <= bytes && (size_t)P.listPoolOff % 8 == 0, "%s: list index", what);
        if (J.screen == 6) EXPECT((size_t)P.exactLdsOffset + 32 * n <= staged, "%s: exact doubles", what);
        for (size_t m = 0; m < (size_t)P.listGroups * P.groupSize; ++m) {
            unsigned short id;
            memcpy(&id, img + P.listShapeOff + 2 * m, 2);
            EXPECT(id == 0xffff || id < n, "%s: member %zu names sphere %u", what, m, id);
        }
        const size_t poolChunks = (bytes - (size_t)P.listPoolOff) / 8;
        EXPECT((n + 3) / 4 <= poolChunks, "%s: list of all spheres", what);
        for (size_t k = 0; k < n * nL; ++k) {
            unsigned w;
            memcpy(&w, img + P.listIndexOff + 4 * k, 4);
            const size_t len = w >> 20, first = w & 0xfffffu;
            if (len == kListAll) continue;
            EXPECT(first + (len + 3) / 4 <= poolChunks, "%s: list %zu runs past the pool", what, k);
            for (size_t e = 0; e < len && first * 4 + e < poolChunks * 4; ++e) {
                unsigned short id;
                memcpy(&id, img + P.listPoolOff + 2 * (first * 4 + e), 2);
                EXPECT(id < n && id != k / nL, "%s: list %zu holds sphere %u", what, k, id);
            }
        }
        if (J.screen == 7 && J.mirrors) {
            const size_t cells = (size_t)P.gridNx * P.gridNy * P.gridNz;
            EXPECT(cells >= 1 && cells <= (size_t)kGridMaxCells && (size_t)P.gridCellsOff + 8 * cells <= (size_t)P.gridRefsOff &&
                   (size_t)P.gridRefsOff <= staged, "%s: grid", what);
        }
    }
    (void)S;
}

static void wellFormed(int scenes) {
    TableCache cache;
    int levels[kMaxScreen + 1] = {0};
    unsigned long long imagesFnv = 1469598103934665603ull;  // over every packed image: the same whatever the thread count
    for (int seed = 0; seed < scenes; ++seed) {
        Scene S = randomScene((uint64_t)seed);
        for (int cap = 0; cap <= kMaxScreen; ++cap) {
            const int groups[] = {0, 4, 12, 32};
            Settings set;
            set.screenCap = cap;
            set.groupSize = groups[(seed + cap) % 4];
            PackedJob J;
            std::string err;
            const int rc = packJob(&S.scene, &S.view, set, cache, J, err);
            EXPECT(rc == GRAFT_OK, "seed %d cap %d: %s", seed, cap, err.c_str());
            if (rc != GRAFT_OK) continue;
            EXPECT(J.screen <= cap, "seed %d: level %d above cap %d", seed, J.screen, cap);
            levels[J.screen]++;
            for (unsigned char b : J.image) imagesFnv = (imagesFnv ^ b) * 1099511628211ull;
            char what[64];
            snprintf(what, sizeof what, "seed %d cap %d level %d", seed, cap, J.screen);
            checkStructure(J, S, what);
            std::vector<unsigned char> arena(J.image.size());  // bindImage only does pointer arithmetic
            const TraceParams P = bindImage(J, arena.data());
            EXPECT((const unsigned char *)P.tileCursor >= arena.data() && (const unsigned char *)P.tileCursor < arena.data() + arena.size(), "%s: cursor", what);
            std::vector<unsigned char> key, key2;
            jobKey(&S.scene, &S.view, set, key);
            jobKey(&S.scene, &S.view, set, key2);
            EXPECT(key == key2 && !key.empty(), "%s: job key", what);
        }
    }
    printf("well-formed scenes: %d x %d caps; jobs per level:", scenes, kMaxScreen + 1);
    for (int l = 0; l <= kMaxScreen; ++l) printf(" %d:%d", l, levels[l]);
    printf("\n");
    printf("images fnv %016llx (%d host threads)\n", imagesFnv, hostThreads());
    for (int l = 0; l <= kMaxScreen; ++l)
        if (scenes >= 200) EXPECT(levels[l] > 0, "no job reached level %d", l);
}

static int pack(const Scene &S, std::string *text = nullptr, int cap = kMaxScreen) {
    static TableCache cache;
    Settings set;
    set.screenCap = cap;
    PackedJob J;
    std::string err;
    const int rc = packJob(&S.scene, &S.view, set, cache, J, err);
    if (text) *text = err;
    if (rc == GRAFT_OK) checkStructure(J, S, "hostile");
    return rc;
}

static void hostile() {
    const double inf = std::numeric_limits<double>::infinity(), nan = std::numeric_limits<double>::quiet_NaN();
    Scene base = randomScene(7);
    for (uint64_t seed = 8; base.shapes.size() < 100 || base.shapes.back().kind != GRAFT_SPHERE || base.shapes[0].transparency > 0; ++seed)
        base = randomScene(seed);
    EXPECT(pack(base) == GRAFT_OK, "base scene");
    {
        TableCache cache;
        PackedJob J;
        std::string err;
        EXPECT(packJob(nullptr, &base.view, Settings(), cache, J, err) == GRAFT_ERR_ARG, "null scene");
        EXPECT(packJob(&base.scene, nullptr, Settings(), cache, J, err) == GRAFT_ERR_ARG, "null view");
    }
    {
        Scene S = base; S.bind(); S.scene.numShapes = -1;
        EXPECT(pack(S) == GRAFT_ERR_ARG, "negative shape count");
        S = base; S.bind(); S.scene.numLights = -5;
        EXPECT(pack(S) == GRAFT_ERR_ARG, "negative light count");
        S = base; S.bind(); S.scene.shapes = nullptr;
        EXPECT(pack(S) == GRAFT_ERR_ARG, "null shapes");
        S = base; S.bind(); S.view.colOffsets = nullptr;
        EXPECT(pack(S) == GRAFT_ERR_ARG, "null column offsets");
        S = base; S.bind(); S.view.nRows = 0;
        EXPECT(pack(S) == GRAFT_ERR_ARG, "empty frame");
        S = base; S.bind(); S.view.nCols = -3;
        EXPECT(pack(S) == GRAFT_ERR_ARG, "negative frame");
        S = base; S.bind(); S.shapes[3].kind = 7;
        EXPECT(pack(S) == GRAFT_ERR_ARG, "unknown shape kind");
        S = base; S.bind(); S.scene.maxDepth = 99;
        EXPECT(pack(S) == GRAFT_ERR_UNSUPPORTED, "depth beyond the stack");
        S = base; S.bind(); S.scene.maxDepth = -4;
        EXPECT(pack(S) == GRAFT_OK, "negative depth traces nothing");
        S = base; S.bind(); S.view.aliasing = 1; S.view.numPasses = 0;
        EXPECT(pack(S) == GRAFT_ERR_ARG, "anti-aliasing with no passes");
        S = base; S.bind(); S.view.aliasing = 1; S.view.numPasses = 1025;
        EXPECT(pack(S) == GRAFT_ERR_ARG, "too many passes");
        S = base; S.bind(); S.view.aliasing = 1; S.view.numPasses = 1024;
        EXPECT(pack(S) == GRAFT_OK, "1024 passes");
    }
    {   // avoid slices outside the index array; indices outside the scene are ignored (contains() compares pointers)
        Scene S = base;
        S.lights[0].avoidOffset = 0; S.lights[0].avoidNum = (int32_t)S.avoid.size() + 1;
        S.bind();
        EXPECT(pack(S) == GRAFT_ERR_ARG, "avoid slice past the end");
        S.lights[0].avoidOffset = -1; S.lights[0].avoidNum = 1;
        EXPECT(pack(S) == GRAFT_ERR_ARG, "negative avoid offset");
        S.lights[0].avoidOffset = 2147483647; S.lights[0].avoidNum = 2147483647;
        EXPECT(pack(S) == GRAFT_ERR_ARG, "avoid slice that overflows");
        S.lights[0].avoidOffset = 0; S.lights[0].avoidNum = 0;
        S.avoid = {-7, 1 << 30, 5, 5, (int32_t)S.shapes.size()};
        S.lights[0].avoidNum = 5;
        S.bind();
        EXPECT(pack(S) == GRAFT_OK, "avoid indices outside the scene");
        S.scene.avoidIndex = nullptr;
        EXPECT(pack(S) == GRAFT_ERR_ARG, "null avoid array with a slice");
    }
    {   // coordinates no table can hold: the job must come out at a level that does not need them
        const double bad[] = {nan, inf, -inf, 1e300, -1e300, 1e17, 4e5};
        for (double v : bad) {
            Scene S = base; S.shapes[5].geom[0] = v; S.bind();
            EXPECT(pack(S) == GRAFT_OK, "sphere centre %g", v);
            S = base; S.shapes[5].geom[3] = v; S.bind();
            EXPECT(pack(S) == GRAFT_OK, "sphere radius %g", v);
            S = base; S.lights[0].loc[1] = v; S.bind();
            EXPECT(pack(S) == GRAFT_OK, "light position %g", v);
            S = base; S.view.camPos[2] = v; S.bind();
            EXPECT(pack(S) == GRAFT_OK, "camera position %g", v);
            S = base; S.cols[S.cols.size() / 2] = v; S.bind();
            EXPECT(pack(S) == GRAFT_OK, "column offset %g", v);
            S = base; S.view.subStepX = v; S.view.aliasing = 1; S.view.numPasses = 3; S.bind();
            EXPECT(pack(S) == GRAFT_OK, "sub-pixel step %g", v);
            S = base; S.view.lrVec[0] = v; S.bind();
            EXPECT(pack(S) == GRAFT_OK, "screen axis %g", v);
            S = base; S.shapes[2].color[1] = v; S.shapes[2].reflectivity = v; S.bind();
            EXPECT(pack(S) == GRAFT_OK, "material %g", v);
        }
        Scene S = base;
        for (GraftShape &s : S.shapes) s.geom[3] = 0;
        S.bind();
        EXPECT(pack(S) == GRAFT_OK, "zero radii");
        S = base;
        S.shapes[0].geom[3] = -5; S.shapes[1].geom[3] = 1e-170;
        S.bind();
        EXPECT(pack(S) == GRAFT_OK, "negative and vanishing radii");
        S = base;
        S.view.lrVec[0] = S.view.lrVec[1] = S.view.lrVec[2] = 0;
        S.bind();
        EXPECT(pack(S) == GRAFT_OK, "zero screen axis");
        S = base;
        std::swap(S.cols[1], S.cols[S.cols.size() - 2]);   // offsets out of order
        S.bind();
        EXPECT(pack(S) == GRAFT_OK, "offsets out of order");
    }
    {   // 4096 coincident spheres: every list is as long as it gets, every grid cell overflows
        Scene S = base;
        S.shapes.clear();
        for (int k = 0; k < 4096; ++k) S.shapes.push_back(sphere(10, 20, 150, 25, k % 2 ? 0.5 : 0.0));
        S.avoid.clear();
        for (GraftLight &l : S.lights) l.avoidNum = l.avoidOffset = 0;
        S.bind();
        for (int cap = 0; cap <= kMaxScreen; ++cap) EXPECT(pack(S, nullptr, cap) == GRAFT_OK, "coincident spheres, cap %d", cap);
        for (GraftShape &s : S.shapes) s.reflectivity = 0;
        S.lights[0].avoidOffset = 0; S.lights[0].avoidNum = 2; S.avoid = {1, 4000};
        S.bind();
        for (int cap = 5; cap <= kMaxScreen; ++cap) EXPECT(pack(S, nullptr, cap) == GRAFT_OK, "coincident matte spheres, cap %d", cap);
    }
    {   // a frame of one column, of one pixel, of 70000 columns
        Scene S = base;
        frame(S, 1, 300, 512, GRAFT_PERSPECTIVE, 2); S.bind();
        EXPECT(pack(S) == GRAFT_OK, "one column");
        frame(S, 1, 1, 512, GRAFT_PERSPECTIVE, 0); S.bind();
        EXPECT(pack(S) == GRAFT_OK, "one pixel");
        frame(S, 70000, 3, 512, GRAFT_PERSPECTIVE, 0); S.bind();
        EXPECT(pack(S) == GRAFT_OK, "70000 columns");
    }
}

static void hostSteps() {
    const double inf = std::numeric_limits<double>::infinity(), nan = std::numeric_limits<double>::quiet_NaN();
    double out[16];
    EXPECT(loopOffsets(512, 512, nullptr, 0) == 512, "512 steps");
    EXPECT(loopOffsets(512 * 1.1, 512, nullptr, 0) == 513, "the zoomed loop runs 513 times");
    EXPECT(loopOffsets(512, 8, out, 3) == 8 && out[0] == -256 && out[2] == -128, "cap below the trip count");
    for (double e : {0.0, -1.0, nan, inf, -inf}) EXPECT(loopOffsets(e, 8, out, 16) == -1, "extent %g", e);
    EXPECT(loopOffsets(1e-300, 8, out, 16) > 0 && loopOffsets(1e300, 4, out, 16) == 4, "extreme extents");
    EXPECT(loopOffsets(512, 0, out, 16) == -1 && loopOffsets(512, -3, out, 16) == -1, "no steps");
    EXPECT(loopOffsets(4e-320, 3, out, 16) == -1 || loopOffsets(4e-320, 3, out, 16) > 0, "denormal extent ends");
    EXPECT(screenCoord(0.0, 512, 512) == 256 && screenCoord(-256.0, 512, 512) == 0, "screen coordinates");
    for (double v : {nan, inf, -inf, 1e300}) {
        (void)screenCoord(v, 512, 512);
        (void)screenCoord(1.0, 512, v);
    }
    (void)screenCoord(1.0, 512, 0.0);
    double sc[3], cp[3];
    const double t[3] = {0, 0, 0}, zero[3] = {0, 0, 0}, dir[3] = {0, 0, -1};
    camInitializeHost(t, dir, 256, 1000, sc, cp);
    EXPECT(sc[2] == -256 && cp[2] == -1256, "camInitialize");
    camInitializeHost(t, zero, 256, 1000, sc, cp);  // 0/0: NaNs, as the reference would get
    camInitializeHost(t, dir, -inf, nan, sc, cp);
    for (int rows : {1, 7, 8, 9, 64, 100, 4097, 16384, 2147483647})
        for (int g = 1; g <= 16; ++g)
            for (int per : {-3, 0, 1, 5, 1000000}) {
                const int count = framePlan(rows, g, per, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
                EXPECT(count >= 1, "plan %d rows %d devices", rows, g);
                if (count > 100000) continue;
                std::vector<int> first(count), n(count), owner(count), at(count), owned(g);
                EXPECT(framePlan(rows, g, per, first.data(), n.data(), owner.data(), at.data(), owned.data(), count) == count, "plan twice");
                long long covered = 0;
                std::vector<long long> seen(g, 0);
                for (int k = 0; k < count; ++k) {
                    EXPECT(first[k] == covered && n[k] > 0 && owner[k] >= 0 && owner[k] < g && at[k] == seen[owner[k]], "piece %d", k);
                    covered += n[k];
                    seen[owner[k]] += n[k];
                }
                EXPECT(covered == rows, "pieces cover %lld of %d rows", covered, rows);
                for (int d = 0; d < g; ++d) EXPECT(seen[d] == owned[d], "device %d owns %d", d, owned[d]);
                int small[2] = {-1, -1};
                EXPECT(framePlan(rows, g, per, small, nullptr, nullptr, nullptr, nullptr, 1) == count && small[1] == -1, "cap of one");
            }
    EXPECT(framePlan(0, 4, 2, nullptr, nullptr, nullptr, nullptr, nullptr, 0) == -1 && framePlan(64, 0, 2, nullptr, nullptr, nullptr, nullptr, nullptr, 0) == -1,
           "empty frame or no device");
}

int main(int argc, char **argv) {
    const int scenes = argc > 1 ? atoi(argv[1]) : 240;
    const bool verbose = getenv("HOST_PACK_VERBOSE") != nullptr;
    if (verbose) fprintf(stderr, "host steps\n");
    hostSteps();
    if (verbose) fprintf(stderr, "hostile inputs\n");
    hostile();
    if (verbose) fprintf(stderr, "well-formed scenes\n");
    wellFormed(scenes);
    if (failures) {
        fprintf(stderr, "%d expectation(s) failed\n", failures);
        return 1;
    }
    printf("host pack check ok\n");
    return 0;
}
