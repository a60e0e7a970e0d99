// host_job.h -- host side: turns a GraftScene + GraftView into the image of the one device block a trace kernel
// reads (geometry, screen tables, materials, lights, loop offsets, camera, ticket counters) plus the launch
// parameters that go with it.  No HIP call is made here: the caller decides where the image lives (an allocation
// of its own, a reused buffer, the library's static scratch block).  Included by grafthip.hip only.
#pragma once

#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <string>
#include <vector>

#include "trace_types.h"
#include "pack_tables.h"

namespace graft {

// Diagnostic settings (A/B timing and the tests that run every screen level).  They never change a result.
struct Settings {
    int screenCap = kMaxScreen;  // highest screen level a job may use
    int groupSize = 0;       // spheres per group of the two-level screens, 0 = chosen per scene
    int tilesPerTicket = 0;  // tiles a wave draws at once, 0 = chosen per launch
    int reservedCUs = 0;     // CUs a trace's persistent grid leaves free (for transfer kernels running beside it)
    int copyPieceKB = -1;    // graftRenderRows: a frame leaves for the host in pieces of about this size (0: in one piece, -1: chosen per frame)
};

struct PackedJob {
    TraceParams P;                     // everything but the pointers, which bindImage() fills in
    std::vector<unsigned char> image;  // the device block, byte for byte
    size_t oGeom = 0, oGeomF = 0, oMember = 0, oRuns = 0, oMat = 0, oShade = 0, oLights = 0, oMask = 0, oCol = 0, oRow = 0,
           oCam = 0, oGrid = 0, oCursor = 0;
    int avoidMode = 0;
    int screen = 0;         // 0 = exact only, 1 = screen per sphere, 2 = two-level ball screen, 3/4 = per-family screens
    bool lds = false;       // hot table staged in LDS
    bool exactLds = false;  // exact doubles staged in LDS behind the screen table
    bool transp = false;    // some shape is transparent: tree-walk kernel
    bool mirrors = false;   // some shape has reflectivity > 0
    size_t ldsBytes = 0;
    int block = kSmallBlock;
};

// Tables that are expensive to build and depend only on part of the input, kept from one packJob to the next: an
// interactive host re-traces the same scene after every key press (mainTracing.c:552-631), and a key that changes
// flags, zoom or the view direction leaves the spheres and the lights where they were.  The ball table depends on the
// sphere geometry and the group size; a cone table on those plus its apex and the (rounded-up) coordinate scale.
struct TableCache {
    std::vector<double> sph;
    int groupSize = -1;
    bool haveBalls = false, ballsOk = false;
    TwoLevelTable balls;
    bool haveGrid = false, gridOk = false;
    GridTable grid;
    struct Cone {
        double apex[3], scale;
        TwoLevelTable table;
    };
    std::list<Cone> cones;  // a list: references handed out stay valid while others are added
    struct Ortho {          // camera table of an orthographic view (packOrthoTable)
        double centre[3], dir[3], scale;
        OrthoTable table;
    };
    std::list<Ortho> orthos;
    struct Occluders {      // occluder lists of one light (packLightLists)
        double light[3], scale;
        std::vector<unsigned char> avoid;  // the light's avoid row (empty: nothing avoided)
        bool ok;
        LightLists lists;
    };
    std::list<Occluders> occluders;
    unsigned long long builds = 0;  // tables built so far (tests and the bench read it through the diagnostics getter)

    void reset(const std::vector<double> &s, int g) {
        sph = s;
        groupSize = g;
        haveBalls = ballsOk = false;
        haveGrid = gridOk = false;
        cones.clear();
        orthos.clear();
        occluders.clear();
    }
    void makeRoom(size_t wanted) {  // call before a round of cone() / occluderLists() calls, never between them
        if (cones.size() + wanted > 32) cones.clear();
        if (orthos.size() > 8) orthos.clear();
        if (occluders.size() + wanted > 32) occluders.clear();
    }
    const Occluders &occluderLists(const double light[3], double scale, const unsigned char *avoidRow, size_t n) {
        for (Occluders &o : occluders) {
            if (o.light[0] != light[0] || o.light[1] != light[1] || o.light[2] != light[2] || o.scale != scale) continue;
            bool same = avoidRow ? o.avoid.size() == n && memcmp(o.avoid.data(), avoidRow, n) == 0 : o.avoid.empty();
            if (same) return o;
        }
        occluders.emplace_back();
        Occluders &o = occluders.back();
        o.light[0] = light[0]; o.light[1] = light[1]; o.light[2] = light[2];
        o.scale = scale;
        if (avoidRow) o.avoid.assign(avoidRow, avoidRow + n);
        o.ok = packLightLists(sph, light, scale, avoidRow, o.lists);
        builds++;
        return o;
    }
    const OrthoTable &ortho(const double centre[3], const double dir[3], double scale) {
        for (Ortho &o : orthos)
            if (memcmp(o.centre, centre, sizeof o.centre) == 0 && memcmp(o.dir, dir, sizeof o.dir) == 0 && o.scale == scale) return o.table;
        orthos.emplace_back();
        Ortho &o = orthos.back();
        memcpy(o.centre, centre, sizeof o.centre);
        memcpy(o.dir, dir, sizeof o.dir);
        o.scale = scale;
        packOrthoTable(sph, centre, dir, scale, groupSize, o.table);
        builds++;
        return o.table;
    }
    const TwoLevelTable &cone(const double apex[3], double scale) {
        for (Cone &c : cones)
            if (c.apex[0] == apex[0] && c.apex[1] == apex[1] && c.apex[2] == apex[2] && c.scale == scale) return c.table;
        cones.emplace_back();
        Cone &c = cones.back();
        c.apex[0] = apex[0]; c.apex[1] = apex[1]; c.apex[2] = apex[2];
        c.scale = scale;
        packConeTable(sph, apex, scale, groupSize, c.table);
        builds++;
        return c.table;
    }
};

// Level 5: the cone families' group and third-level sections in LDS as at level 4 (member sections in global memory),
// no ball family; rays outside the cone families walk a uniform grid whose cells, entries and per-sphere screen table
// are staged behind the third-level sections.  `fam` = cone tables then the ball table (whose entries only pad the
// interleaved table's stride here); `inter` = the interleaved table of all of them.  Returns false when the grid
// cannot be built or the sections do not fit.
inline bool packLevel5(TableCache &cache, const std::vector<double> &sph, size_t cones,
                       const std::vector<const TwoLevelTable *> &fam, std::vector<float> &inter,
                       std::vector<int> &interShape, TraceParams &P) {
    if (!cache.haveGrid) {
        cache.gridOk = packGrid(sph, cache.grid);
        cache.haveGrid = true;
        cache.builds++;
    }
    if (!cache.gridOk || cones == 0) return false;
    const GridTable &G = cache.grid;
    // the kernel reads blocks of family f at stride `families`: re-interleave the cone families alone
    std::vector<const TwoLevelTable *> conesOnly(fam.begin(), fam.begin() + cones);
    std::vector<float> table, supers;
    std::vector<int> shapes;
    interleaveFamilies(conesOnly, table, shapes);
    interleaveSupers(conesOnly, supers);
    const size_t headFloats = 4 * (size_t)conesOnly[0]->memberBase * cones;
    const size_t words = headFloats + supers.size() + G.cells.size() + G.refs.size() / 2 + G.aos.size();
    if (sizeof(float) * words > kLdsMax) return false;
    P.ldsHeadFloats = (int)headFloats;
    P.ballMembersLds = (int)(headFloats / 4);  // where the tail starts (float4 index)
    P.ldsTailSrc = (int)table.size();
    P.superLds = (int)(headFloats / 4);
    P.superEntries = conesOnly[0]->superEntries;
    P.gridNx = G.n[0]; P.gridNy = G.n[1]; P.gridNz = G.n[2];
    P.gridCellsOff = (int)(sizeof(float) * (headFloats + supers.size()));
    P.gridRefsOff = P.gridCellsOff + (int)(sizeof(unsigned) * G.cells.size());
    P.gridAosOff = (P.gridRefsOff + (int)(sizeof(unsigned short) * G.refs.size()) + 15) & ~15;
    const size_t tailWords = (size_t)(P.gridAosOff / 4) - headFloats + G.aos.size();
    P.ldsTailFloats = (int)tailWords;
    std::vector<float> tail(tailWords, 0.0f);
    memcpy(tail.data(), supers.data(), sizeof(float) * supers.size());
    memcpy((char *)tail.data() + (P.gridCellsOff - sizeof(float) * headFloats), G.cells.data(), sizeof(unsigned) * G.cells.size());
    memcpy((char *)tail.data() + (P.gridRefsOff - sizeof(float) * headFloats), G.refs.data(), sizeof(unsigned short) * G.refs.size());
    memcpy((char *)tail.data() + (P.gridAosOff - sizeof(float) * headFloats), G.aos.data(), sizeof(float) * G.aos.size());
    if (sizeof(float) * (headFloats + tailWords) > kLdsMax) return false;
    table.insert(table.end(), tail.begin(), tail.end());
    inter.swap(table);
    interShape.swap(shapes);
    return true;
}

// Levels 6 and 7 (closestHitLists): every primary ray looks only at the spheres its TILE can see (a list the wave builds
// per tile from the camera's two-level cone table), every shadow ray only at the spheres that can lie between the
// sphere it starts on and its light (lists built here, per light).  Builds the image described at
// TraceParams::listGroups and sets the launch parameters; returns false when the job does not qualify or the image
// does not fit.  `inLds`: the whole image is staged in LDS, exact doubles included (level 6); otherwise only the
// spheres' screen table and -- `wantGrid`: the scene has mirrors, whose rays walk it -- the uniform grid are, and the
// lists and the camera table stay in global memory behind them (level 7).  `staged` = bytes of the image that go to LDS.
inline bool packLists(TableCache &cache, const std::vector<double> &sph, const GraftView *view,
                      const std::vector<DevLight> &lights, const std::vector<unsigned char> &mask, int nS, double scale,
                      bool inLds, bool wantGrid, std::vector<float> &image, TraceParams &P, size_t &staged,
                      size_t &ldsBytes, int &block) {
    const size_t n = sph.size() / 4, nL = lights.size();
    if (n > 65535 || n * nL > (1u << 22)) return false;
    // the tile's bounding cone is drawn around the rays of its first and last columns and rows
    for (int i = 1; i < view->nCols; ++i)
        if (!(view->colOffsets[i] >= view->colOffsets[i - 1])) return false;
    for (int i = 1; i < view->nRows; ++i)
        if (!(view->rowOffsets[i] >= view->rowOffsets[i - 1])) return false;
    if (wantGrid) {
        if (!cache.haveGrid) {
            cache.gridOk = packGrid(sph, cache.grid);
            cache.haveGrid = true;
            cache.builds++;
        }
        if (!cache.gridOk) return false;
    }
    const GridTable &grid = cache.grid;
    // the camera table: cones seen from the camera, or -- orthographic view -- balls in the plane across the rays
    const bool ortho = view->projectionType == GRAFT_ORTHOGRAPHIC;
    const TwoLevelTable *camTable = nullptr;
    const OrthoTable *orthoTable = nullptr;
    if (ortho) {
        const double vv = view->viewDir[0] * view->viewDir[0] + view->viewDir[1] * view->viewDir[1] + view->viewDir[2] * view->viewDir[2];
        if (!(vv > 1e-200 && vv < 1e200)) return false;
        const double len = sqrt(vv);
        const double dir[3] = {-view->viewDir[0] / len, -view->viewDir[1] / len, -view->viewDir[2] / len};  // mainTracing.c:410-413
        orthoTable = &cache.ortho(view->screenCenter, dir, scale);
    } else {
        camTable = &cache.cone(view->camPos, scale);
    }
    const std::vector<int> &camShape = ortho ? orthoTable->memberShape : camTable->memberShape;
    const size_t gs = (size_t)(ortho ? orthoTable->kGroup : camTable->kGroup), G = camShape.size() / gs;
    std::vector<const LightLists *> lists;
    for (size_t i = 0; i < nL; ++i) {
        const unsigned char *row = lights[i].avoidNum > 0 ? mask.data() + i * (size_t)nS : nullptr;
        const TableCache::Occluders &o = cache.occluderLists(lights[i].loc, scale, row, n);
        if (!o.ok) return false;
        lists.push_back(&o.lists);
    }
    auto up = [](size_t x, size_t a) { return (x + a - 1) / a * a; };
    const size_t entries = G + G * gs;  // group entries, then member entries
    const size_t oAos = 0, oExact = up(oAos + 16 * n, 32), afterExact = inLds ? oExact + 32 * n : oAos + 16 * n;
    const size_t oCells = up(afterExact, 16), oRefs = oCells + (wantGrid ? 4 * grid.cells.size() : 0);
    const size_t afterGrid = up(oRefs + (wantGrid ? 2 * grid.refs.size() : 0), 16);
    // (the camera table's group entries and their sines first: they are scanned for every tile list and are staged in
    // LDS at level 7 as well; the member entries, read for the few groups a tile opens, follow)
    const size_t oGroups = afterGrid, oSines = oGroups + 16 * G, afterGroups = up(oSines + 4 * G, 16);
    const size_t oMembers = afterGroups, oMemberSines = oMembers + 16 * G * gs, oShape = oMemberSines + 4 * G * gs,
                 oIndex = up(oShape + 2 * G * gs, 4), oPool = up(oIndex + 4 * n * nL, 8);
    size_t poolEntries = up(n, 4);  // the list of all spheres comes first; every list starts at a multiple of 4 entries
    std::vector<size_t> base(nL);
    for (size_t i = 0; i < nL; ++i) { base[i] = poolEntries; poolEntries += lists[i]->pool.size(); }
    if (poolEntries / 4 >= (1u << 20)) return false;
    const size_t blob = up(oPool + 2 * poolEntries, 16);
    staged = inLds ? blob : afterGroups;
    // in LDS behind the staged part: a tile list per wave and, where it fits, the waves' tables of sample scale factors
    // (launches of fewer than four samples per pixel run kernels that do not look at the table: launchTrace)
    const bool wantSamples = view->aliasing && view->numPasses >= 2 && view->numPasses <= kSampleTablePasses;
    const size_t perWave = 2 * kTileListCap, perWaveSamples = 8 * 16 * (size_t)(wantSamples ? view->numPasses : 0);
    block = 0;
    P.sampleTableOff = -1;
    for (int withSamples = wantSamples ? 1 : 0; withSamples >= 0 && !block; --withSamples)
        for (int b = kSmallBlock; b <= kLargeBlock; b *= 2) {
            const size_t waves = (size_t)(b / 64);
            const size_t total = staged + waves * perWave + (withSamples ? waves * perWaveSamples : 0);
            const size_t limit = b == kSmallBlock ? kLdsSmall : b == 2 * kSmallBlock ? 2 * kLdsSmall : kLdsMax;
            if (total <= limit) {
                block = b;
                ldsBytes = total;
                if (withSamples) P.sampleTableOff = (int)(staged + waves * perWave);
                break;
            }
        }
    if (!block) return false;
    std::vector<unsigned char> img(blob, 0);
    for (size_t k = 0; k < n; ++k) {
        float e[4] = {(float)sph[4 * k], (float)sph[4 * k + 1], (float)sph[4 * k + 2], 0.0f};
        if (!screenRho(&sph[4 * k], sph[4 * k + 3], e[3])) return false;
        memcpy(&img[oAos + 16 * k], e, sizeof e);
    }
    if (inLds) memcpy(&img[oExact], sph.data(), 32 * n);
    if (wantGrid) {
        memcpy(&img[oCells], grid.cells.data(), 4 * grid.cells.size());
        memcpy(&img[oRefs], grid.refs.data(), 2 * grid.refs.size());
    }
    // cone entries as {axis, cos A'} and sin A', from the table's -kappa2 = -(cos^2 A'): the cosine rounded down, the
    // sine up (either way the cone only widens); an entry that always passes gets cos = -2, one that never does +inf
    auto coneEntries = [&](size_t first, size_t count, size_t oEntries, size_t oSin) {
        std::vector<float> plain;
        unpackEntries(camTable->entries, first, count, plain);
        for (size_t e = 0; e < count; ++e) {
            const double k2 = -(double)plain[4 * e + 3];
            float c = INFINITY, sn = 0.0f;
            if (k2 < 0) {
                c = -2.0f;
            } else if (k2 <= 1.0) {
                c = floatBelow(sqrt(k2) * (1.0 - 1e-12));
                sn = -floatBelow(-sqrt(1.0 - (double)c * (double)c) * (1.0 + 1e-12));
            }
            plain[4 * e + 3] = c;
            memcpy(&img[oEntries + 16 * e], &plain[4 * e], 16);
            memcpy(&img[oSin + 4 * e], &sn, 4);
        }
    };
    if (ortho) {  // {q, R} entries as they are; the sines' slots stay zero
        memcpy(&img[oGroups], orthoTable->groups.data(), 16 * G);
        memcpy(&img[oMembers], orthoTable->members.data(), 16 * G * gs);
    } else {
        coneEntries(0, G, oGroups, oSines);
        coneEntries((size_t)camTable->memberBase, G * gs, oMembers, oMemberSines);
    }
    for (size_t m = 0; m < G * gs; ++m) {
        const unsigned short id = camShape[m] < 0 ? (unsigned short)0xffff : (unsigned short)camShape[m];
        memcpy(&img[oShape + 2 * m], &id, 2);
    }
    for (size_t k = 0; k < n; ++k)
        for (size_t i = 0; i < nL; ++i) {
            unsigned w = lists[i]->index[k];
            if ((w >> 20) != kListAll) w += (unsigned)(base[i] / 4);  // (chunk numbers stay below 2^20: checked above)
            memcpy(&img[oIndex + 4 * (k * nL + i)], &w, 4);
        }
    for (size_t k = 0; k < n; ++k) {
        const unsigned short id = (unsigned short)k;
        memcpy(&img[oPool + 2 * k], &id, 2);
    }
    for (size_t i = 0; i < nL; ++i)
        if (!lists[i]->pool.empty()) memcpy(&img[oPool + 2 * base[i]], lists[i]->pool.data(), 2 * lists[i]->pool.size());
    image.assign(blob / 4, 0.0f);
    memcpy(image.data(), img.data(), blob);
    P.exactLdsOffset = (int)oExact;
    P.listGroups = (int)G;
    P.listGroupsOff = (int)oGroups; P.listSinesOff = (int)oSines; P.listShapeOff = (int)oShape;
    P.listMembersOff = (int)oMembers; P.listMemberSinesOff = (int)oMemberSines;
    P.listIndexOff = (int)oIndex; P.listPoolOff = (int)oPool; P.listTileOff = (int)staged;
    if (wantGrid) {
        P.gridNx = grid.n[0]; P.gridNy = grid.n[1]; P.gridNz = grid.n[2];
        P.gridCellsOff = (int)oCells; P.gridRefsOff = (int)oRefs; P.gridAosOff = (int)oAos;
    }
    return true;
}

// Validate, pack shapes to per-kind geometry arrays + materials, avoid lists to indices, and build the screen
// tables.  Replaces the implicit "read the globals" of launchRays (mainTracing.c:387-391).  Returns a GRAFT_* code
// and, on failure, a message in `err`.
inline int packJob(const GraftScene *scene, const GraftView *view, const Settings &set, TableCache &cache,
                   PackedJob &J, std::string &err) {
    char msg[256];
    auto fail = [&](int code, const char *fmt, auto... args) {
        snprintf(msg, sizeof msg, fmt, args...);
        err = msg;
        return code;
    };
    if (!scene || !view) return fail(GRAFT_ERR_ARG, "%s", "null scene or view");
    const int nS = scene->numShapes, nL = scene->numLights;
    if (nS < 0 || nL < 0 || (nS > 0 && !scene->shapes) || (nL > 0 && !scene->lights))
        return fail(GRAFT_ERR_ARG, "bad shape/light arrays (%d shapes, %d lights)", nS, nL);
    if (view->nCols <= 0 || view->nRows <= 0 || !view->colOffsets || !view->rowOffsets)
        return fail(GRAFT_ERR_ARG, "bad frame %d x %d", view->nCols, view->nRows);
    if (view->aliasing && view->numPasses < 1)
        return fail(GRAFT_ERR_ARG, "numPasses %d with anti-aliasing on", view->numPasses);
    if (view->aliasing && view->numPasses > 1024) return fail(GRAFT_ERR_ARG, "numPasses %d too large", view->numPasses);
    if (scene->maxDepth > kMaxDepth) return fail(GRAFT_ERR_UNSUPPORTED, "maxDepth %d > %d", scene->maxDepth, kMaxDepth);

    bool transparent = false, mirrors = false;
    std::vector<DevRun> runs;
    std::vector<DevMaterial> mats(nS);
    std::vector<double> sph, pln;
    for (int k = 0; k < nS; ++k) {
        const GraftShape &s = scene->shapes[k];
        if (s.kind != GRAFT_SPHERE && s.kind != GRAFT_PLANE) return fail(GRAFT_ERR_ARG, "shape %d has kind %d", k, s.kind);
        if (s.transparency > 0) transparent = true;
        if (s.reflectivity > 0) mirrors = true;
        int slot;
        if (s.kind == GRAFT_SPHERE) {
            slot = (int)(sph.size() / 4);
            sph.insert(sph.end(), {s.geom[0], s.geom[1], s.geom[2], s.geom[3] * s.geom[3]});  // shapes.c:54 radius*radius
        } else {
            slot = (int)(pln.size() / 6);
            pln.insert(pln.end(), s.geom, s.geom + 6);
        }
        if (runs.empty() || runs.back().kind != s.kind) runs.push_back({s.kind, k, 0, slot, 0, 0, {0, 0}});
        runs.back().count++;
        DevMaterial &m = mats[k];
        for (int c = 0; c < 3; ++c) m.color[c] = s.color[c];
        m.reflectivity = s.reflectivity; m.ambientLight = s.ambientLight;
        m.transparency = s.transparency; m.refrIndex = s.refrIndex;
        m.slot = slot; m.kind = s.kind;
    }
    // avoid lists: contains() (mainTracing.c:116-123) is set membership by pointer identity; entries that
    // name no shape of the scene can never match
    int avoidMode = 0;
    std::vector<DevLight> lights(nL);
    std::vector<unsigned char> mask((size_t)nL * (nS > 0 ? nS : 1) + 1, 0);
    for (int i = 0; i < nL; ++i) {
        const GraftLight &l = scene->lights[i];
        if (l.avoidNum < 0 || l.avoidOffset < 0 || (long long)l.avoidOffset + l.avoidNum > scene->avoidIndexLen ||
            (l.avoidNum > 0 && !scene->avoidIndex))
            return fail(GRAFT_ERR_ARG, "light %d avoid slice [%d,+%d) outside %d", i, l.avoidOffset, l.avoidNum,
                        scene->avoidIndexLen);
        DevLight &d = lights[i];
        memset(&d, 0, sizeof d);
        for (int c = 0; c < 3; ++c) { d.color[c] = l.color[c]; d.loc[c] = l.loc[c]; }
        d.avoidFirst = -1;
        for (int a = 0; a < l.avoidNum; ++a) {
            const int s = scene->avoidIndex[l.avoidOffset + a];
            if (s < 0 || s >= nS || mask[(size_t)i * nS + s]) continue;
            mask[(size_t)i * nS + s] = 1;
            if (d.avoidNum++ == 0) d.avoidFirst = s;
        }
        if (d.avoidNum == 1 && avoidMode < 1) avoidMode = 1;
        if (d.avoidNum > 1) avoidMode = 2;
    }

    TraceParams &P = J.P;
    memset(&P, 0, sizeof P);
    DevCamera cam;
    memset(&cam, 0, sizeof cam);
    for (int c = 0; c < 3; ++c) {
        cam.screenCenter[c] = view->screenCenter[c]; cam.camPos[c] = view->camPos[c]; cam.viewDir[c] = view->viewDir[c];
        cam.lrVec[c] = view->lrVec[c]; cam.udVec[c] = view->udVec[c]; cam.background[c] = scene->backgroundColor[c];
    }
    cam.lrDen = cam.lrVec[0] * cam.lrVec[0] + cam.lrVec[1] * cam.lrVec[1] + cam.lrVec[2] * cam.lrVec[2];
    cam.udDen = cam.udVec[0] * cam.udVec[0] + cam.udVec[1] * cam.udVec[1] + cam.udVec[2] * cam.udVec[2];
    cam.subStepX = view->subStepX; cam.subStepY = view->subStepY;
    for (int c = 0; c < 3; ++c) {
        cam.toScreen[c] = cam.screenCenter[c] - cam.camPos[c];
        cam.lrUnit[c] = cam.lrVec[c] / sqrt(cam.lrDen);
        cam.udUnit[c] = cam.udVec[c] / sqrt(cam.udDen);
    }
    if (view->projectionType == GRAFT_ORTHOGRAPHIC) {
        // orthographic tile lists work in the plane across the rays: the unit vectors along lrVec and udVec with their
        // components along the rays' direction taken out (zero for the reference's own views, whose three vectors are
        // perpendicular; any other view is covered as well)
        const double vv = cam.viewDir[0] * cam.viewDir[0] + cam.viewDir[1] * cam.viewDir[1] + cam.viewDir[2] * cam.viewDir[2];
        if (vv > 1e-200 && vv < 1e200) {
            const double len = sqrt(vv);
            const double dir[3] = {-cam.viewDir[0] / len, -cam.viewDir[1] / len, -cam.viewDir[2] / len};
            const double lrAlong = cam.lrUnit[0] * dir[0] + cam.lrUnit[1] * dir[1] + cam.lrUnit[2] * dir[2];
            const double udAlong = cam.udUnit[0] * dir[0] + cam.udUnit[1] * dir[1] + cam.udUnit[2] * dir[2];
            for (int c = 0; c < 3; ++c) {
                cam.lrUnit[c] -= lrAlong * dir[c];
                cam.udUnit[c] -= udAlong * dir[c];
            }
        }
    }
    P.projectionType = view->projectionType; P.aliasing = view->aliasing ? 1 : 0;
    P.numPasses = view->numPasses; P.maxDepth = scene->maxDepth;
    P.numShapes = nS; P.numLights = nL; P.numRuns = (int)runs.size(); P.numSpheres = (int)(sph.size() / 4);
    P.nCols = view->nCols; P.nRows = view->nRows;
    P.geomDoubles = (int)(sph.size() + pln.size());
    P.avoidTotal = 0;
    for (int i = 0; i < nL; ++i) P.avoidTotal += lights[i].avoidNum;
    {   // largest magnitudes of light and shape colours: their products must stay finite
        double maxLight = 0, maxShape = 0;
        bool finite = true;
        for (int i = 0; i < nL; ++i)
            for (int c = 0; c < 3; ++c) {
                finite = finite && std::isfinite(lights[i].color[c]);
                maxLight = fmax(maxLight, fabs(lights[i].color[c]));
            }
        for (int k = 0; k < nS; ++k) {
            const DevMaterial &m = mats[k];
            for (int c = 0; c < 3; ++c) {
                finite = finite && std::isfinite(m.color[c]);
                maxShape = fmax(maxShape, fabs(m.color[c]));
            }
            finite = finite && std::isfinite(m.reflectivity) && std::isfinite(m.ambientLight) && std::isfinite(m.transparency);
        }
        // Small scenes keep asking every light down to the first unshadowed one: there the closest-hit query is
        // cheap next to the shading code, and lanes that drop a query fall out of step with their wave, which then
        // runs the code of every stage in every round (start-up scene, 11 shapes: 3.98 instead of 2.70 ms).
        P.skipLightsBehind = finite && maxLight < 1e150 && maxShape < 1e150 && nS >= kSkipBehindMinShapes ? 1 : 0;
    }
    {   // numPasses <= 1024 (checked above), so samples < 2^20 and e = ceil(2^32 / n) n - 2^32 < n gives sample e < 2^30
        const unsigned long long n = P.numPasses > 0 ? (unsigned long long)P.numPasses : 1;
        const unsigned long long recip = ((1ull << 32) + n - 1) / n;
        P.numPassesRecip = recip > 0xffffffffull ? 0xffffffffu : (unsigned)recip;  // (n = 1: the only sample is 0)
    }
    P.fusedMiss = 1;
    for (size_t k = 0; k < sph.size() / 4; ++k)
        if (sph[4 * k + 3] < 1e-280) P.fusedMiss = 0;

    const int screenCap = set.screenCap;
    int screen = (screenCap >= 1 && !sph.empty()) ? 1 : 0;
    if (transparent) avoidMode = 2;  // the tree-walk kernels exist for the general avoid mode only
    std::vector<float> tableF;
    std::vector<int> memberShape;
    // Spheres per group of the two-level screens.  Larger groups halve the scan over the group entries at the
    // price of more member tests per opened group: measured on 1024 spheres, 16 per group is 17 % faster than 8
    // (12: +11 %, 24: +14 %, 32: +11 %), on 256 spheres there is no difference, and the three-level walk of scenes
    // too large for LDS (members in global memory) is fastest with 8.
    int groupSize = kGroup;
    {
        const size_t n = sph.size() / 4;
        const size_t F = nL <= 7 ? (size_t)nL + (view->projectionType == GRAFT_PERSPECTIVE ? 1 : 0) + 1 : 1;
        const size_t nGroups = (n + 15) / 16, groupEntries = (nGroups + 7) / 8 * 8;
        if (n >= 512 && 16 * F * (groupEntries + 8 + nGroups * 16 + 8) <= kLdsMax) groupSize = 16;
    }
    if (set.groupSize >= 4 && set.groupSize <= 64 && set.groupSize % 4 == 0) groupSize = set.groupSize;
    P.groupSize = groupSize;
    const bool wantGroups = screen && screenCap >= 2 && pln.empty() && !transparent;
    // the apexes of the ray families: the camera (perspective only), then every light
    std::vector<std::array<double, 3>> apexes;
    if (view->projectionType == GRAFT_PERSPECTIVE) apexes.push_back({view->camPos[0], view->camPos[1], view->camPos[2]});
    const int lightBase = (int)apexes.size();
    for (int i = 0; i < nL; ++i) apexes.push_back({lights[i].loc[0], lights[i].loc[1], lights[i].loc[2]});
    // scale: the largest |.|_1 of anything a screened ray can start from or aim at, rounded up to a power of two
    // (any larger value is valid: it only widens the margins) so that zooming or turning the view leaves the
    // lights' tables reusable
    double scale = 0;
    {
        const double offs = fabs(view->subStepX) * (view->numPasses + 1) + fabs(view->subStepY) * (view->numPasses + 1);
        double colMax = 0, rowMax = 0;
        for (int i = 0; i < view->nCols; ++i) colMax = fmax(colMax, fabs(view->colOffsets[i]));
        for (int i = 0; i < view->nRows; ++i) rowMax = fmax(rowMax, fabs(view->rowOffsets[i]));
        scale = fabs(view->screenCenter[0]) + fabs(view->screenCenter[1]) + fabs(view->screenCenter[2]) + 2 * (colMax + rowMax + offs);
        for (size_t k = 0; k < sph.size() / 4; ++k)
            scale = fmax(scale, fabs(sph[4 * k]) + fabs(sph[4 * k + 1]) + fabs(sph[4 * k + 2]) + sqrt(sph[4 * k + 3]));
        for (const auto &a : apexes) scale = fmax(scale, fabs(a[0]) + fabs(a[1]) + fabs(a[2]));
        if (scale > 0 && scale < kGroupScale) scale = fmax(1.0, exp2(ceil(log2(scale))));  // (NaN stays NaN)
    }
    // Candidate lists first: they need neither the ball table nor the lights' cone tables.  Level 6: everything in LDS
    // (scenes without mirrors that fit); level 7: lists in global memory, and a grid in LDS for the rays behind a mirror.
    bool spheresInRange = sph.size() / 4 >= 64;
    for (size_t k = 0; k < sph.size() / 4 && spheresInRange; ++k)  // (what packBallTable asks of a scene; rejects NaN)
        spheresInRange = fabs(sph[4 * k]) + fabs(sph[4 * k + 1]) + fabs(sph[4 * k + 2]) + sqrt(sph[4 * k + 3]) < kGroupScale;
    // (up to 31 lights: the lists need no table per light in LDS, only an index word per sphere and light)
    if (wantGroups && screenCap >= 6 && nL <= 31 && scale < kGroupScale && spheresInRange) {
        if (cache.groupSize != groupSize || cache.sph != sph) cache.reset(sph, groupSize);
        cache.makeRoom(apexes.size());
        std::vector<float> image;
        size_t staged = 0, ldsBytes = 0;
        int block = 0, got = 0;
        if (!mirrors && packLists(cache, sph, view, lights, mask, nS, scale, true, false, image, P, staged, ldsBytes, block)) got = 6;
        else if (screenCap >= 7 && packLists(cache, sph, view, lights, mask, nS, scale, false, mirrors, image, P, staged, ldsBytes, block)) got = 7;
        if (got) {
            screen = got;
            tableF.swap(image);
            P.familyScale = scale;
            P.ldsHeadFloats = (int)(staged / 4);
            J.ldsBytes = ldsBytes;
            J.block = block;
        }
    }
    const TwoLevelTable *balls = nullptr;
    if (wantGroups && screen < 6) {
        if (cache.groupSize != groupSize || cache.sph != sph) cache.reset(sph, groupSize);
        if (!cache.haveBalls) {
            cache.ballsOk = packBallTable(sph, groupSize, cache.balls);
            cache.haveBalls = true;
            cache.builds++;
        }
        if (cache.ballsOk) balls = &cache.balls;
    }
    if (balls) {
        screen = 2;
        tableF = balls->entries;
        memberShape = balls->memberShape;
        P.groupEntries = balls->groupEntries;
        P.memberBase = balls->memberBase;
    }
    // per-family tables: cones seen from the camera (perspective only) and from every light, then the ball table
    if (screen == 2 && screenCap >= 3 && nL <= 7) {
        if (!apexes.empty() && scale < kGroupScale) {  // (NaN fails the comparison)
            std::vector<const TwoLevelTable *> fam;
            cache.makeRoom(apexes.size());
            for (size_t f = 0; f < apexes.size(); ++f) fam.push_back(&cache.cone(apexes[f].data(), scale));
            fam.push_back(balls);
            std::vector<float> inter;
            std::vector<int> interShape;
            interleaveFamilies(fam, inter, interShape);
            const size_t headFloats = 4 * (size_t)balls->memberBase * fam.size();
            const size_t ballMemberFloats = balls->entries.size() - 4 * (size_t)balls->memberBase;
            std::vector<float> supers;
            interleaveSupers(fam, supers);
            if (sizeof(float) * inter.size() <= kLdsMax) {
                screen = 3;
            } else if (screenCap >= 5 && packLevel5(cache, sph, apexes.size(), fam, inter, interShape, P)) {
                screen = 5;
            } else if (screenCap >= 4 && sizeof(float) * (headFloats + ballMemberFloats + supers.size()) <= kLdsMax) {
                // cone member sections stay in global memory; the ball family's member section and the third-level
                // sections are appended to the table so that the kernel can stage them behind the group sections
                screen = 4;
                P.ldsHeadFloats = (int)headFloats;
                P.ballMembersLds = (int)(headFloats / 4);
                P.ldsTailSrc = (int)inter.size();
                P.ldsTailFloats = (int)(ballMemberFloats + supers.size());
                P.superLds = (int)((headFloats + ballMemberFloats) / 4);
                P.superEntries = balls->superEntries;
                inter.insert(inter.end(), balls->entries.begin() + 4 * (size_t)balls->memberBase, balls->entries.end());
                inter.insert(inter.end(), supers.begin(), supers.end());
            }
            if (screen >= 3) {
                tableF.swap(inter);
                memberShape.swap(interShape);
                P.families = (int)fam.size();
                P.ballFamily = screen == 5 ? -1 : (int)fam.size() - 1;
                if (screen == 5) P.families = (int)fam.size() - 1;
                P.cameraFamily = lightBase ? 0 : -1;
                P.lightFamilyBase = lightBase;
                P.familyScale = scale;
            }
        }
    }
    if (screen < 2) { P.groupEntries = 0; P.memberBase = 0; memberShape.clear(); }
    if (screen == 1) packScreenTable(sph, runs, tableF);
    P.filterFloats = (int)tableF.size();

    // one block: geometry | float table | member shapes | runs | materials | lights | mask | colOff | rowOff | camera |
    // ticket counters.  The geometry is followed by kGeomPad doubles of padding: the exact sphere loop reads one
    // block of four spheres ahead.
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    J.oGeom = 0;
    J.oGeomF = align(J.oGeom + sizeof(double) * (P.geomDoubles + kGeomPad));
    J.oMember = align(J.oGeomF + sizeof(float) * (tableF.size() + 1));
    J.oRuns = align(J.oMember + sizeof(int) * (memberShape.size() + 1));
    J.oMat = align(J.oRuns + sizeof(DevRun) * (runs.size() + 1));
    J.oShade = align(J.oMat + sizeof(DevMaterial) * (nS + 1));
    J.oLights = align(J.oShade + sizeof(DevShade) * (nS + 1));
    J.oMask = align(J.oLights + sizeof(DevLight) * (nL + 1));
    J.oCol = align(J.oMask + mask.size() + 1);
    J.oRow = align(J.oCol + sizeof(double) * view->nCols);
    J.oCam = align(J.oRow + sizeof(double) * view->nRows);
    J.oGrid = align(J.oCam + sizeof(DevCamera));
    J.oCursor = align(J.oGrid + sizeof(DevGrid));
    const size_t total = J.oCursor + sizeof(int) * kCursors * kCursorStride + 256;
    std::vector<unsigned char> &host = J.image;
    host.assign(total, 0);
    if (!sph.empty()) memcpy(&host[J.oGeom], sph.data(), sizeof(double) * sph.size());
    if (!pln.empty()) memcpy(&host[J.oGeom + sizeof(double) * sph.size()], pln.data(), sizeof(double) * pln.size());
    // padding spheres that every ray misses: a large negative FINITE r*r, so that the discriminant is a negative finite
    // number and the block test of the exact loop (is any of four values not negative and finite?) passes over them
    for (int i = 0; i < kGeomPad; i += 4) {
        const double never[4] = {0, 0, 0, -1e300};
        memcpy(&host[J.oGeom + sizeof(double) * (P.geomDoubles + i)], never, sizeof never);
    }
    if (!tableF.empty()) memcpy(&host[J.oGeomF], tableF.data(), sizeof(float) * tableF.size());
    if (!memberShape.empty()) memcpy(&host[J.oMember], memberShape.data(), sizeof(int) * memberShape.size());
    if (!runs.empty()) memcpy(&host[J.oRuns], runs.data(), sizeof(DevRun) * runs.size());
    if (nS) memcpy(&host[J.oMat], mats.data(), sizeof(DevMaterial) * nS);
    for (int k = 0; k < nS; ++k) {  // (this file is compiled without contraction of multiply and add, like the kernels)
        const DevMaterial &m = mats[k];
        DevShade sh;
        for (int c = 0; c < 3; ++c) {
            sh.color[c] = m.color[c];
            const double amb = m.ambientLight * m.color[c];
            sh.ambient[c] = amb + 0.0;
        }
        sh.wDiff = (1 - m.ambientLight) * (1 - m.reflectivity) * (1 - m.transparency);
        const double wRefl = (1 - m.ambientLight) * m.reflectivity;
        sh.mirrorZero = wRefl * 0.0;
        memcpy(&host[J.oShade + sizeof(DevShade) * k], &sh, sizeof sh);
    }
    if (nL) memcpy(&host[J.oLights], lights.data(), sizeof(DevLight) * nL);
    if (!mask.empty()) memcpy(&host[J.oMask], mask.data(), mask.size());
    memcpy(&host[J.oCol], view->colOffsets, sizeof(double) * view->nCols);
    memcpy(&host[J.oRow], view->rowOffsets, sizeof(double) * view->nRows);
    memcpy(&host[J.oCam], &cam, sizeof cam);
    if (screen == 5 || (screen == 7 && mirrors)) {
        DevGrid g;
        for (int a = 0; a < 3; ++a) { g.lo[a] = cache.grid.lo[a]; g.hi[a] = cache.grid.hi[a]; g.cs[a] = cache.grid.cs[a]; g.invCs[a] = cache.grid.invCs[a]; }
        memcpy(&host[J.oGrid], &g, sizeof g);
    }

    J.avoidMode = avoidMode;
    J.screen = screen;
    J.mirrors = mirrors;
    const size_t tableBytes = screen >= 4 ? sizeof(float) * ((size_t)P.ldsHeadFloats + P.ldsTailFloats)
                            : screen ? sizeof(float) * tableF.size() : sizeof(double) * (size_t)(P.geomDoubles + kGeomPad);
    const size_t exactBytes = sizeof(double) * (size_t)P.geomDoubles;
    J.transp = transparent;
    if (screen >= 6) {  // (LDS image, workgroup size and the offset of the exact doubles: packLists)
        J.lds = true;
        J.exactLds = false;
        return GRAFT_OK;
    }
    J.lds = tableBytes <= kLdsMax;
    J.exactLds = !transparent && J.lds && screen && screen < 4 && ((tableBytes + 31) & ~(size_t)31) + exactBytes <= kLdsMax;
    P.exactLdsOffset = (int)((tableBytes + 31) & ~(size_t)31);
    J.ldsBytes = !J.lds ? 0 : J.exactLds ? P.exactLdsOffset + exactBytes : tableBytes;
    // workgroup size: as many workgroups per CU as the LDS footprint allows, 16 waves per CU in every case
    J.block = J.ldsBytes <= kLdsSmall ? kSmallBlock : J.ldsBytes <= 2 * kLdsSmall ? 2 * kSmallBlock : kLargeBlock;
    return GRAFT_OK;
}

// the launch parameters of a packed job whose image has been copied to device address `base`
inline TraceParams bindImage(const PackedJob &J, void *base) {
    TraceParams P = J.P;
    unsigned char *b = (unsigned char *)base;
    P.geom = (const double *)(b + J.oGeom);
    P.geomF = (const float *)(b + J.oGeomF);
    P.memberShape = (const int *)(b + J.oMember);
    P.runs = (const DevRun *)(b + J.oRuns);
    P.materials = (const DevMaterial *)(b + J.oMat);
    P.shade = (const DevShade *)(b + J.oShade);
    P.lights = (const DevLight *)(b + J.oLights);
    P.avoidMask = (const unsigned char *)(b + J.oMask);
    P.colOff = (const double *)(b + J.oCol);
    P.rowOff = (const double *)(b + J.oRow);
    P.camera = (const DevCamera *)(b + J.oCam);
    P.grid = (const DevGrid *)(b + J.oGrid);
    P.tileCursor = (int *)(b + J.oCursor);
    return P;
}

// Everything packJob reads, as bytes: two calls with equal keys produce equal jobs.
inline void jobKey(const GraftScene *scene, const GraftView *view, const Settings &set, std::vector<unsigned char> &key) {
    key.clear();
    auto put = [&](const void *p, size_t n) {
        const unsigned char *b = (const unsigned char *)p;
        key.insert(key.end(), b, b + n);
    };
    put(&set, sizeof set);
    put(&scene->numShapes, sizeof(int32_t)); put(&scene->numLights, sizeof(int32_t));
    put(&scene->avoidIndexLen, sizeof(int32_t)); put(&scene->maxDepth, sizeof(int32_t));
    put(scene->backgroundColor, sizeof scene->backgroundColor);
    if (scene->numShapes > 0 && scene->shapes) put(scene->shapes, sizeof(GraftShape) * (size_t)scene->numShapes);
    if (scene->numLights > 0 && scene->lights) put(scene->lights, sizeof(GraftLight) * (size_t)scene->numLights);
    if (scene->avoidIndexLen > 0 && scene->avoidIndex) put(scene->avoidIndex, sizeof(int32_t) * (size_t)scene->avoidIndexLen);
    put(view, offsetof(GraftView, colOffsets));
    if (view->nCols > 0 && view->colOffsets) put(view->colOffsets, sizeof(double) * (size_t)view->nCols);
    if (view->nRows > 0 && view->rowOffsets) put(view->rowOffsets, sizeof(double) * (size_t)view->nRows);
}

}  // namespace graft
