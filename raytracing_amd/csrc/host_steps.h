// host_steps.h -- the steps of launchRays that stay on the host (loop headers, screen coordinates, camInitialize)
// and the plan of a frame's pieces over several GPUs: pure host arithmetic, no HIP call.  Included by grafthip.hip,
// whose extern "C" entry points are these functions, and by the host-only sanitizer build (tests/host_asan).
#pragma once

#include <cmath>
#include <vector>

#include "trace_types.h"

namespace graft {

// the values of `for (double i = -extent/2; i < extent/2; i = i + extent/n)` (mainTracing.c:392-393)
inline int loopOffsets(double extent, int n, double *out, int cap) {
    if (!(extent > 0) || !std::isfinite(extent) || n <= 0) return -1;
    int count = 0;
    for (double i = -extent / 2; i < extent / 2; i = i + extent / n) {  // mainTracing.c:392
        if (out && count < cap) out[count] = i;
        if (++count > 2 * n + 16) return -1;
    }
    return count;
}

inline int screenCoord(double coord, int n, double extent) {  // mainTracing.c:89-100
    double ratio = n / extent;
    coord = round(coord * ratio);
    // (the reference converts whatever comes out; a value no int holds -- a zero extent, a NaN -- has no pixel)
    const double shifted = coord + n / 2;
    if (!(shifted > -2147483648.0 && shifted < 2147483648.0)) return -2147483647 - 1;
    return (int)shifted;
}

inline void newPointHost(const double start[3], const double dir[3], double dist, double out[3]) {
    double den = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];  // mainTracing.c:107
    double scalar = dist < 0 ? -sqrt(dist * dist / den) : sqrt(dist * dist / den);
    for (int c = 0; c < 3; ++c) out[c] = start[c] + scalar * dir[c];
}

inline void camInitializeHost(const double target[3], const double viewDir[3], double targetToScreen, double screenToCam,
                              double screenCenter[3], double camPos[3]) {
    double sc[3];
    newPointHost(target, viewDir, targetToScreen, sc);  // mainTracing.c:381
    newPointHost(sc, viewDir, screenToCam, camPos);     // mainTracing.c:383
    for (int c = 0; c < 3; ++c) screenCenter[c] = sc[c];
}

// The pieces of a frame of nRows rows over nDevices GPUs, piecesPerDevice each (fewer when the frame has fewer rows of
// tiles): piece k covers rows [firstRow[k], firstRow[k] + rows[k]), belongs to GPU owner[k] = k mod nDevices and starts
// at row rowInOwner[k] of that GPU's own buffer, which holds ownedRows[owner[k]] rows in all.  Writes at most `cap`
// entries per array (any array may be null); returns the number of pieces, -1 for a frame or device count below 1.
inline int framePlan(int nRows, int nDevices, int piecesPerDevice, int *firstRow, int *rows, int *owner, int *rowInOwner,
                     int *ownedRows, int cap) {
    if (nRows <= 0 || nDevices <= 0) return -1;
    const int tiles = (int)(((long long)nRows + kTile - 1) / kTile);
    long long want = (long long)nDevices * (piecesPerDevice > 1 ? piecesPerDevice : 1);
    int count = (int)(want > tiles ? tiles : want);
    if (count < 1) count = 1;
    std::vector<int> owned((size_t)nDevices, 0);
    int out = 0;
    for (int k = 0; k < count; ++k) {
        const long long begin = (long long)k * tiles / count * kTile, end = (long long)(k + 1) * tiles / count * kTile;
        const int r0 = (int)begin, r1 = (int)(end > nRows ? nRows : end);  // (begin is below nRows: a whole tile row lies behind it)
        if (r1 <= r0) continue;
        const int who = k % nDevices;
        if (out < cap) {
            if (firstRow) firstRow[out] = r0;
            if (rows) rows[out] = r1 - r0;
            if (owner) owner[out] = who;
            if (rowInOwner) rowInOwner[out] = owned[who];
        }
        owned[who] += r1 - r0;
        ++out;
    }
    if (ownedRows)
        for (int d = 0; d < nDevices; ++d) ownedRows[d] = owned[d];
    return out;
}

}  // namespace graft
