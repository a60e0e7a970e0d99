// pack_tables.h -- host side: builds the single-precision screen tables the kernels stage in LDS
// (included by grafthip.hip only).
#pragma once

#include <algorithm>
#include <cmath>
#include <utility>
#include <vector>

#include "trace_types.h"

namespace graft {

// Single-precision screen table.  Per sphere: (float)cx, (float)cy, (float)cz and rho >= r*r + the sphere's share
// of the screen's worst-case error (64 u |c|_1^2 + 2.2 u rho, u = 2^-24), rounded up; centre 0 and rho = +inf
// (always passes) for spheres whose coordinates are not finite or are too large for float arithmetic.  Every
// sphere run gets its own block, padded to a multiple of eight with entries that never pass (rho = -inf), and the
// table ends with eight more such entries because the hot loop fetches two blocks of four ahead.  Block layout:
// GRAFT_SCREEN_PACKED = 0: {cx cy cz rho} x 4;  1: {cx0 cx1 cy0 cy1}{cz0 cz1 rho0 rho1}{cx2 cx3 cy2 cy3}{cz2 cz3 rho2 rho3}.
void packScreenTable(const std::vector<double> &sph, std::vector<DevRun> &runs, std::vector<float> &out) {
    const double u = 5.9604644775390625e-08;
    out.clear();
    auto entry = [&](float *blk, int i, float cx, float cy, float cz, float rho) {
#if GRAFT_SCREEN_PACKED
        float *pair = blk + 8 * (i >> 1);
        const int h = i & 1;
        pair[0 + h] = cx; pair[2 + h] = cy; pair[4 + h] = cz; pair[6 + h] = rho;
#else
        blk[4 * i] = cx; blk[4 * i + 1] = cy; blk[4 * i + 2] = cz; blk[4 * i + 3] = rho;
#endif
    };
    for (DevRun &run : runs) {
        if (run.kind != GRAFT_SPHERE) continue;
        run.fslot = (int)(out.size() / 4);
        run.fcount = (run.count + 7) / 8 * 8;
        out.resize(out.size() + 4 * (size_t)run.fcount, 0.0f);
        for (int j = 0; j < run.fcount; ++j) {
            float *blk = out.data() + 4 * (size_t)run.fslot + 16 * (j / 4);
            if (j >= run.count) { entry(blk, j % 4, 0.0f, 0.0f, 0.0f, -INFINITY); continue; }
            const double *c = &sph[4 * (size_t)(run.slot + j)];
            const double c1 = fabs(c[0]) + fabs(c[1]) + fabs(c[2]);
            if (c1 < 1e15 && c[3] >= 0 && c[3] < 1e30) {
                const double want = ((c[3] + 64.0 * u * c1 * c1) / (1.0 - 2.2 * u)) * 1.000001 + 1e-30;
                float rho = (float)want;
                if ((double)rho < want) rho = nextafterf(rho, INFINITY);
                entry(blk, j % 4, (float)c[0], (float)c[1], (float)c[2], rho);
            } else {
                entry(blk, j % 4, 0.0f, 0.0f, 0.0f, INFINITY);
            }
        }
    }
    const size_t body = out.size();
    out.resize(body + 32, 0.0f);
    for (int i = 0; i < 8; ++i) entry(out.data() + body + 16 * (i / 4), i % 4, 0.0f, 0.0f, 0.0f, -INFINITY);
}

// rho of the screen table for a ball of squared radius rr centred at c (see packScreenTable); false if the
// coordinates are not finite or too large for float arithmetic
bool screenRho(const double c[3], double rr, float &rho) {
    const double u = 5.9604644775390625e-08;
    const double c1 = fabs(c[0]) + fabs(c[1]) + fabs(c[2]);
    if (!(c1 < 1e15 && rr >= 0 && rr < 1e30)) return false;
    const double want = ((rr + 64.0 * u * c1 * c1) / (1.0 - 2.2 * u)) * 1.000001 + 1e-30;
    rho = (float)want;
    if ((double)rho < want) rho = nextafterf(rho, INFINITY);
    return true;
}

void putEntry(float *blk, int i, float cx, float cy, float cz, float rho) {
#if GRAFT_SCREEN_PACKED
    float *pair = blk + 8 * (i >> 1);
    const int h = i & 1;
    pair[0 + h] = cx; pair[2 + h] = cy; pair[4 + h] = cz; pair[6 + h] = rho;
#else
    blk[4 * i] = cx; blk[4 * i + 1] = cy; blk[4 * i + 2] = cz; blk[4 * i + 3] = rho;
#endif
}

constexpr int kGroup = 8;          // spheres per group of the two-level screen
constexpr double kGroupMargin = 0.1;
constexpr double kGroupScale = 1e5;  // coordinates must stay below this for the margin to cover the rounding terms

// Two-level screen table for a scene of spheres only (see closestHitGrouped).  Layout, in entries of 4 floats:
// [group table: bounding balls, 4 per block, padded to a multiple of 8][8 entries of padding: the loop fetches two
// blocks ahead][member table: 8 entries (2 blocks) per group].  Returns false when the scene does not qualify.
bool packGroupedTable(const std::vector<double> &sph, std::vector<float> &out, std::vector<int> &memberShape,
                      int &groupEntries, int &memberBase) {
    const int n = (int)(sph.size() / 4);
    if (n < 64) return false;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < n; ++k) {
        const double *c = &sph[4 * (size_t)k];
        const double r = sqrt(c[3]);
        if (!(fabs(c[0]) + fabs(c[1]) + fabs(c[2]) + r < kGroupScale)) return false;  // also rejects NaN
        for (int a = 0; a < 3; ++a) { lo[a] = fmin(lo[a], c[a]); hi[a] = fmax(hi[a], c[a]); }
    }
    // groups = leaves of a k-d tree: split the longest axis of the centres' bounding box at the median (left part a
    // multiple of the group size) until at most kGroup spheres remain; ties by index, so the result is deterministic
    std::vector<std::pair<unsigned, int>> order(n);
    {
        std::vector<int> idx(n);
        for (int k = 0; k < n; ++k) idx[k] = k;
        std::vector<std::pair<int, int>> todo{{0, n}};
        while (!todo.empty()) {
            const std::pair<int, int> seg = todo.back();
            todo.pop_back();
            const int count = seg.second - seg.first;
            if (count <= kGroup) continue;
            double blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int i = seg.first; i < seg.second; ++i)
                for (int a = 0; a < 3; ++a) {
                    blo[a] = fmin(blo[a], sph[4 * (size_t)idx[i] + a]);
                    bhi[a] = fmax(bhi[a], sph[4 * (size_t)idx[i] + a]);
                }
            int axis = 0;
            for (int a = 1; a < 3; ++a)
                if (bhi[a] - blo[a] > bhi[axis] - blo[axis]) axis = a;
            const int half = (count / 2 + kGroup - 1) / kGroup * kGroup;
            std::sort(idx.begin() + seg.first, idx.begin() + seg.second, [&](int x, int y) {
                const double cx = sph[4 * (size_t)x + axis], cy = sph[4 * (size_t)y + axis];
                return cx < cy || (cx == cy && x < y);
            });
            todo.push_back({seg.first, seg.first + half});
            todo.push_back({seg.first + half, seg.second});
        }
        for (int k = 0; k < n; ++k) order[k] = {0u, idx[k]};
    }
    const int nGroups = (n + kGroup - 1) / kGroup;
    groupEntries = (nGroups + 7) / 8 * 8;
    memberBase = groupEntries + 8;
    out.assign(4 * ((size_t)memberBase + (size_t)nGroups * kGroup + 8), 0.0f);
    memberShape.assign((size_t)nGroups * kGroup, -1);
    for (int i = 0; i < memberBase; ++i) putEntry(out.data() + 16 * (i / 4), i % 4, 0.0f, 0.0f, 0.0f, -INFINITY);
    for (int gI = 0; gI < nGroups; ++gI) {
        const int first = gI * kGroup, last = std::min(n, first + kGroup);
        // centre of the bounding ball: start at the centroid and walk towards the farthest sphere with shrinking
        // steps (Badoiu-Clarkson); any centre is valid, the radius below is what makes the ball enclose the members
        double g[3] = {0, 0, 0};
        for (int m = first; m < last; ++m)
            for (int a = 0; a < 3; ++a) g[a] += sph[4 * (size_t)order[m].second + a] / (last - first);
        for (int it = 0; it < 64; ++it) {
            int far = first;
            double farDist = -1;
            for (int m = first; m < last; ++m) {
                const double *c = &sph[4 * (size_t)order[m].second];
                const double dx = c[0] - g[0], dy = c[1] - g[1], dz = c[2] - g[2];
                const double dd = sqrt(dx * dx + dy * dy + dz * dz) + sqrt(c[3]);
                if (dd > farDist) { farDist = dd; far = m; }
            }
            const double *c = &sph[4 * (size_t)order[far].second];
            for (int a = 0; a < 3; ++a) g[a] += (c[a] - g[a]) / (it + 2);
        }
        double R = 0;
        for (int m = first; m < last; ++m) {
            const double *c = &sph[4 * (size_t)order[m].second];
            const double dx = c[0] - g[0], dy = c[1] - g[1], dz = c[2] - g[2];
            R = fmax(R, sqrt(dx * dx + dy * dy + dz * dz) + sqrt(c[3]));
        }
        R = R * (1.0 + 1e-9) + kGroupMargin;
        float rho;
        if (!screenRho(g, R * R * (1.0 + 1e-12), rho)) return false;
        putEntry(out.data() + 16 * (gI / 4), gI % 4, (float)g[0], (float)g[1], (float)g[2], rho);
        float *mblk = out.data() + 4 * ((size_t)memberBase + (size_t)gI * kGroup);
        for (int m = 0; m < kGroup; ++m) {
            float *blk = mblk + 16 * (m / 4);
            if (first + m >= last) { putEntry(blk, m % 4, 0.0f, 0.0f, 0.0f, -INFINITY); continue; }
            const int k = order[first + m].second;
            const double *c = &sph[4 * (size_t)k];
            if (!screenRho(c, c[3], rho)) return false;
            putEntry(blk, m % 4, (float)c[0], (float)c[1], (float)c[2], rho);
            memberShape[(size_t)gI * kGroup + m] = k;
        }
    }
    const size_t tail = (size_t)memberBase + (size_t)nGroups * kGroup;
    for (int i = 0; i < 8; ++i) putEntry(out.data() + 4 * tail + 16 * (i / 4), i % 4, 0.0f, 0.0f, 0.0f, -INFINITY);
    return true;
}


}  // namespace graft
