// pack_tables.h -- host side: builds the single-precision screen tables the kernels stage in LDS
// (included by grafthip.hip only).
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

#include "trace_types.h"

namespace graft {

// The tables of a large scene are built on several host threads: body(begin, end) runs over contiguous ranges of
// [0, n), at least `grain` items each, on at most hostThreads() threads (the calling thread takes the first range).  What
// a body computes for an item must not depend on the split: the tables are the same whatever the thread count.
inline int hostThreads() {
    static const int threads = [] {
        int t = (int)std::thread::hardware_concurrency();
        if (const char *e = getenv("GRAFT_HOST_THREADS"))
            if (atoi(e) > 0) t = atoi(e);
        return std::max(1, std::min(t, 16));
    }();
    return threads;
}

template <class Body>
inline void parallelRanges(int n, int grain, Body body) {
    const int t = std::min(hostThreads(), n / std::max(1, grain));
    if (t <= 1) {
        body(0, n);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve((size_t)t - 1);
    auto range = [&body, n, t](int i) { body((int)((long long)n * i / t), (int)((long long)n * (i + 1) / t)); };
    int started = 1;  // (range 0 is the calling thread's)
    try {
        for (; started < t; ++started) pool.emplace_back(range, started);
    } catch (const std::system_error &) {  // no more threads to be had: the calling thread takes the ranges that are left
    }
    range(0);
    for (int i = started; i < t; ++i) range(i);
    for (std::thread &th : pool) th.join();
}

// Single-precision screen table.  Per sphere: (float)cx, (float)cy, (float)cz and rho >= r*r + the sphere's share
// of the screen's worst-case error (64 u |c|_1^2 + 2.2 u rho, u = 2^-24), rounded up; centre 0 and rho = +inf
// (always passes) for spheres whose coordinates are not finite or are too large for float arithmetic.  Every
// sphere run gets its own block, padded to a multiple of eight with entries that never pass (rho = -inf), and the
// table ends with eight more such entries because the hot loop fetches two blocks of four ahead.  Block layout:
// {cx0 cx1 cy0 cy1}{cz0 cz1 rho0 rho1}{cx2 cx3 cy2 cy3}{cz2 cz3 rho2 rho3}.
inline void packScreenTable(const std::vector<double> &sph, std::vector<DevRun> &runs, std::vector<float> &out) {
    const double u = 5.9604644775390625e-08;
    out.clear();
    auto entry = [&](float *blk, int i, float cx, float cy, float cz, float rho) {
        float *pair = blk + 8 * (i >> 1);
        const int h = i & 1;
        pair[0 + h] = cx; pair[2 + h] = cy; pair[4 + h] = cz; pair[6 + h] = rho;
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
inline bool screenRho(const double c[3], double rr, float &rho) {
    const double u = 5.9604644775390625e-08;
    const double c1 = fabs(c[0]) + fabs(c[1]) + fabs(c[2]);
    if (!(c1 < 1e15 && rr >= 0 && rr < 1e30)) return false;
    const double want = ((rr + 64.0 * u * c1 * c1) / (1.0 - 2.2 * u)) * 1.000001 + 1e-30;
    rho = (float)want;
    if ((double)rho < want) rho = nextafterf(rho, INFINITY);
    return true;
}

inline void putEntry(float *blk, int i, float cx, float cy, float cz, float rho) {
    float *pair = blk + 8 * (i >> 1);
    const int h = i & 1;
    pair[0 + h] = cx; pair[2 + h] = cy; pair[4 + h] = cz; pair[6 + h] = rho;
}

constexpr int kGroup = 8;          // default spheres per group of the two-level screens (a multiple of 4)
constexpr int kSuper = 8;          // groups per entry of the third level
constexpr double kGroupMargin = 0.1;
constexpr double kGroupScale = 1e5;  // coordinates must stay below this for the margin to cover the rounding terms

// Order of n points (3 doubles each) such that every run of kGroup consecutive points is a leaf of a k-d tree, and
// every aligned run of kGroup * kSuper points a subtree: split the longest axis of the bounding box at the median
// (left part a multiple of kGroup * kSuper while more than that many points remain, of kGroup below) until at most
// kGroup points remain; ties by index, so the result is deterministic.
inline std::vector<int> kdLeafOrder(const std::vector<double> &pts, int n, int kGroup) {
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
                blo[a] = fmin(blo[a], pts[3 * (size_t)idx[i] + a]);
                bhi[a] = fmax(bhi[a], pts[3 * (size_t)idx[i] + a]);
            }
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (bhi[a] - blo[a] > bhi[axis] - blo[axis]) axis = a;
        const int unit = count > kGroup * kSuper ? kGroup * kSuper : kGroup;
        const int half = (count / 2 + unit - 1) / unit * unit;
        std::sort(idx.begin() + seg.first, idx.begin() + seg.second, [&](int x, int y) {
            const double cx = pts[3 * (size_t)x + axis], cy = pts[3 * (size_t)y + axis];
            return cx < cy || (cx == cy && x < y);
        });
        todo.push_back({seg.first, seg.first + half});
        todo.push_back({seg.first + half, seg.second});
    }
    return idx;
}

// One two-level table: [group entries, padded to a multiple of 8][8 entries of padding: the loop fetches two blocks
// ahead][kGroup member entries (kGroup / 4 blocks) per group][8 entries of padding], 4 floats per entry, packed in
// blocks of 4 (putEntry); memberShape[g * kGroup + m] is the shape index of a member entry, -1 for padding.
struct TwoLevelTable {
    std::vector<float> entries;
    std::vector<int> memberShape;
    // third level: one entry per kSuper consecutive groups, padded to a multiple of 8 plus 8 entries (the scan fetches
    // two blocks ahead), same packing
    std::vector<float> supers;
    int groupEntries = 0, memberBase = 0, superEntries = 0, kGroup = graft::kGroup;
    void shape(int n, float neverPass, int groupSize) {
        kGroup = groupSize;
        const int nGroups = (n + kGroup - 1) / kGroup;
        groupEntries = (nGroups + 7) / 8 * 8;
        superEntries = (groupEntries / kSuper + 7) / 8 * 8;
        supers.assign(4 * (size_t)(superEntries + 8), 0.0f);
        for (int i = 0; i < superEntries + 8; ++i) putEntry(supers.data() + 16 * (i / 4), i % 4, 0.0f, 0.0f, 0.0f, neverPass);
        memberBase = groupEntries + 8;
        const size_t total = (size_t)memberBase + (size_t)nGroups * kGroup + 8;
        entries.assign(4 * total, 0.0f);
        memberShape.assign((size_t)nGroups * kGroup, -1);
        for (size_t i = 0; i < total; ++i) putEntry(entries.data() + 16 * (i / 4), (int)(i % 4), 0.0f, 0.0f, 0.0f, neverPass);
    }
    void group(int g, float x, float y, float z, float w) { putEntry(entries.data() + 16 * (g / 4), g % 4, x, y, z, w); }
    void super(int s, float x, float y, float z, float w) { putEntry(supers.data() + 16 * (s / 4), s % 4, x, y, z, w); }
    void member(int g, int m, int shapeIndex, float x, float y, float z, float w) {
        const size_t e = (size_t)memberBase + (size_t)g * kGroup + m;
        putEntry(entries.data() + 16 * (e / 4), (int)(e % 4), x, y, z, w);
        memberShape[(size_t)g * kGroup + m] = shapeIndex;
    }
};

// Two-level BALL table for a scene of spheres only (see closestHitGrouped): groups are k-d leaves of the centres,
// each with a bounding ball.  Returns false when the scene does not qualify.
inline bool packBallTable(const std::vector<double> &sph, int kGroup, TwoLevelTable &T) {
    const int n = (int)(sph.size() / 4);
    if (n < 64) return false;
    std::vector<double> pts(3 * (size_t)n);
    for (int k = 0; k < n; ++k) {
        const double *c = &sph[4 * (size_t)k];
        if (!(fabs(c[0]) + fabs(c[1]) + fabs(c[2]) + sqrt(c[3]) < kGroupScale)) return false;  // also rejects NaN
        for (int a = 0; a < 3; ++a) pts[3 * (size_t)k + a] = c[a];
    }
    const std::vector<int> order = kdLeafOrder(pts, n, kGroup);
    T.shape(n, -INFINITY, kGroup);  // rho = -inf never passes
    // bounding ball of the spheres order[first..last): start at the centroid and walk towards the farthest sphere
    // with shrinking steps (Badoiu-Clarkson); any centre is valid, the radius is what makes the ball enclose them
    auto enclose = [&](int first, int last, double g[3], float &rho) {
        g[0] = g[1] = g[2] = 0;
        for (int m = first; m < last; ++m)
            for (int a = 0; a < 3; ++a) g[a] += sph[4 * (size_t)order[m] + a] / (last - first);
        for (int it = 0; it < 64; ++it) {
            int far = first;
            double farDist = -1;
            for (int m = first; m < last; ++m) {
                const double *c = &sph[4 * (size_t)order[m]];
                const double dx = c[0] - g[0], dy = c[1] - g[1], dz = c[2] - g[2];
                const double dd = sqrt(dx * dx + dy * dy + dz * dz) + sqrt(c[3]);
                if (dd > farDist) { farDist = dd; far = m; }
            }
            const double *c = &sph[4 * (size_t)order[far]];
            for (int a = 0; a < 3; ++a) g[a] += (c[a] - g[a]) / (it + 2);
        }
        double R = 0;
        for (int m = first; m < last; ++m) {
            const double *c = &sph[4 * (size_t)order[m]];
            const double dx = c[0] - g[0], dy = c[1] - g[1], dz = c[2] - g[2];
            R = fmax(R, sqrt(dx * dx + dy * dy + dz * dz) + sqrt(c[3]));
        }
        R = R * (1.0 + 1e-9) + kGroupMargin;
        return screenRho(g, R * R * (1.0 + 1e-12), rho);
    };
    const int nGroups = (n + kGroup - 1) / kGroup;
    for (int gI = 0; gI < nGroups; ++gI) {
        const int first = gI * kGroup, last = std::min(n, first + kGroup);
        double g[3];
        float rho;
        if (!enclose(first, last, g, rho)) return false;
        T.group(gI, (float)g[0], (float)g[1], (float)g[2], rho);
        for (int m = first; m < last; ++m) {
            const int k = order[m];
            const double *c = &sph[4 * (size_t)k];
            if (!screenRho(c, c[3], rho)) return false;
            T.member(gI, m - first, k, (float)c[0], (float)c[1], (float)c[2], rho);
        }
    }
    for (int sI = 0; sI * kSuper < nGroups; ++sI) {
        const int first = sI * kSuper * kGroup, last = std::min(n, first + kSuper * kGroup);
        double g[3];
        float rho;
        if (!enclose(first, last, g, rho)) return false;
        T.super(sI, (float)g[0], (float)g[1], (float)g[2], rho);
    }
    return true;
}

// Largest float <= x
inline float floatBelow(double x) {
    float f = (float)x;
    if ((double)f > x) f = nextafterf(f, -INFINITY);
    return f;
}

// Two-level CONE table for the rays that pass through one fixed point `apex` (the camera position for primary
// rays, a light's position for its shadow rays; see closestHitFamilies).  Seen from the apex, sphere k is a cone
// of half-angle alpha_k = asin(r_k / |c_k - apex|) around u_k = (c_k - apex)/|c_k - apex|; a line through the apex
// with unit direction d meets the sphere iff |u_k . d| >= cos(alpha_k).  Groups are k-d leaves of the directions
// u_k, each with an enclosing cone (axis a, half-angle theta >= angle(a, u_k) + alpha_k).  An entry holds
// (a or u, -kappa2) and passes iff (a . d)^2 - kappa2 >= 0, evaluated in single precision; kappa2 is
// cos^2(angle + margin) - 16 u, rounded down, where
//   - 16 u (u = 2^-24) covers rounding a and d to float and the float evaluation (|error of the dot| <= 5.1 u),
//   - margin = 1.01 sqrt(eBound) / distance makes the exact discriminant of every culled sphere smaller than
//     -eBound, and eBound = 1e-13 (4 S)^2 exceeds, for coordinates below S, the sum of: the reference's own rounding
//     of the discriminant (< 3.3e-15 (|s|_1 + |c|_1)^2), the direction's deviation from unit length times b^2, and
//     the distance (a few ulps of the coordinates) by which a computed ray misses the apex.
// kappa2 = -1 always passes (apex inside or near the sphere, or a cone wider than 80 degrees); +inf never passes.
inline void packConeTable(const std::vector<double> &sph, const double apex[3], double scale, int kGroup,
                          TwoLevelTable &T) {
    const int n = (int)(sph.size() / 4);
    const double u = 5.9604644775390625e-08;
    const double eBound = 1e-13 * (4 * scale) * (4 * scale);
    const double kWide = 1.4;  // radians; wider cones always pass
    std::vector<double> dir(3 * (size_t)n), alpha(n), dist(n);
    for (int k = 0; k < n; ++k) {
        const double *c = &sph[4 * (size_t)k];
        const double w[3] = {c[0] - apex[0], c[1] - apex[1], c[2] - apex[2]};
        const double len = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        const double r = sqrt(c[3]);
        dist[k] = len;
        if (len > r * (1 + 1e-9) + 1e-9 * scale) {
            alpha[k] = asin(r / len);
            for (int a = 0; a < 3; ++a) dir[3 * (size_t)k + a] = w[a] / len;
        } else {  // the apex is inside (or on) the sphere: every line through it meets the sphere
            alpha[k] = M_PI;
            dir[3 * (size_t)k] = 1; dir[3 * (size_t)k + 1] = 0; dir[3 * (size_t)k + 2] = 0;
            if (len > 0) for (int a = 0; a < 3; ++a) dir[3 * (size_t)k + a] = w[a] / len;
        }
    }
    auto angle = [](const double *p, const double *q) {  // accurate for small angles
        const double cx = p[1] * q[2] - p[2] * q[1], cy = p[2] * q[0] - p[0] * q[2], cz = p[0] * q[1] - p[1] * q[0];
        return atan2(sqrt(cx * cx + cy * cy + cz * cz), p[0] * q[0] + p[1] * q[1] + p[2] * q[2]);
    };
    auto kappa2 = [&](double ang, double nearest) -> float {
        const double margin = 1.01 * sqrt(eBound) / nearest + 1e-7;
        if (!(ang + margin < kWide)) return -1.0f;
        const double c = cos(ang + margin);
        return floatBelow(c * c - 16 * u);
    };
    const std::vector<int> order = kdLeafOrder(dir, n, kGroup);
    T.shape(n, -INFINITY, kGroup);  // the table stores -kappa2: -inf never passes
    // enclosing cone of the spheres order[first..last): axis a, and -kappa2 of its half-angle
    auto enclose = [&](int first, int last, double a[3]) -> float {
        a[0] = a[1] = a[2] = 0;
        for (int m = first; m < last; ++m)
            for (int c = 0; c < 3; ++c) a[c] += dir[3 * (size_t)order[m] + c];
        auto normalise = [&]() {
            const double len = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
            if (len > 1e-12) { a[0] /= len; a[1] /= len; a[2] /= len; }
            else { a[0] = 1; a[1] = 0; a[2] = 0; }
        };
        normalise();
        for (int it = 0; it < 64; ++it) {  // walk the axis towards the member that sticks out most
            int far = first;
            double farAng = -1;
            for (int m = first; m < last; ++m) {
                const double t = angle(a, &dir[3 * (size_t)order[m]]) + alpha[order[m]];
                if (t > farAng) { farAng = t; far = m; }
            }
            for (int c = 0; c < 3; ++c) a[c] += (dir[3 * (size_t)order[far] + c] - a[c]) / (it + 2);
            normalise();
        }
        double theta = 0, nearest = INFINITY;
        for (int m = first; m < last; ++m) {
            theta = fmax(theta, angle(a, &dir[3 * (size_t)order[m]]) + alpha[order[m]]);
            nearest = fmin(nearest, dist[order[m]]);
        }
        return -kappa2(theta, nearest);
    };
    const int nGroups = (n + kGroup - 1) / kGroup;
    // (every group and every third-level entry writes floats of its own: ranges of them go to different threads)
    parallelRanges(nGroups, 16, [&](int gBegin, int gEnd) {
        for (int gI = gBegin; gI < gEnd; ++gI) {
            const int first = gI * kGroup, last = std::min(n, first + kGroup);
            double a[3];
            const float w = enclose(first, last, a);
            T.group(gI, (float)a[0], (float)a[1], (float)a[2], w);
            for (int m = first; m < last; ++m) {
                const int k = order[m];
                T.member(gI, m - first, k, (float)dir[3 * (size_t)k], (float)dir[3 * (size_t)k + 1],
                         (float)dir[3 * (size_t)k + 2], -kappa2(alpha[k], dist[k]));
            }
        }
    });
    const int nSupers = (nGroups + kSuper - 1) / kSuper;
    parallelRanges(nSupers, 2, [&](int sBegin, int sEnd) {
        for (int sI = sBegin; sI < sEnd; ++sI) {
            const int first = sI * kSuper * kGroup, last = std::min(n, first + kSuper * kGroup);
            double a[3];
            const float w = enclose(first, last, a);
            T.super(sI, (float)a[0], (float)a[1], (float)a[2], w);
        }
    });
}

// Uniform grid over the spheres for rays that pass through no fixed point (mirror rays and the shadow rays of their
// hits): see closestHitGrid.  Every sphere is entered in every cell that its ball, grown by a margin, touches; a
// point of a sphere's surface therefore lies in a cell that lists the sphere, and so does every point within the
// margin of it -- which is what lets the walk step from cell to cell in single precision: the margin is
// kGroupMargin + 1e-4 S for coordinates up to S <= kGridScale, against a worst-case drift of the stepped ray
// parameter of 4e-5 S (192 steps of half an ulp of a parameter below 3.5 S).  About one cell per
// sphere, at most kGridMaxCells cells and 255 entries per cell.  A cell is two words: {first << 8 | entries, its first
// two sphere numbers (16 bits each)}; refs[first ...] holds the cell's THIRD and later entries, so that the walk reaches
// the spheres of most cells with one read less; aos[] = {cx, cy, cz, rho} per sphere in single precision, rho as in
// packScreenTable.
constexpr int kGridMaxCells = 4096;
constexpr double kGridScale = 32768;
struct GridTable {
    double lo[3], hi[3], cs[3], invCs[3];
    int n[3];
    std::vector<unsigned> cells;
    std::vector<unsigned short> refs;
    std::vector<float> aos;
};

inline bool packGrid(const std::vector<double> &sph, GridTable &T) {
    const int n = (int)(sph.size() / 4);
    if (n < 64 || n > 65535) return false;
    std::vector<double> rad(n);
    double scale = 0;
    for (int k = 0; k < n; ++k) {
        const double *c = &sph[4 * (size_t)k];
        const double s1 = fabs(c[0]) + fabs(c[1]) + fabs(c[2]) + sqrt(c[3]);
        if (!(s1 < kGridScale)) return false;  // also rejects NaN
        scale = fmax(scale, s1);
    }
    const double margin = kGroupMargin + 1e-4 * scale;
    for (int a = 0; a < 3; ++a) { T.lo[a] = INFINITY; T.hi[a] = -INFINITY; }
    for (int k = 0; k < n; ++k) {
        const double *c = &sph[4 * (size_t)k];
        rad[k] = sqrt(c[3]) * (1.0 + 1e-9) + margin;
        for (int a = 0; a < 3; ++a) {
            T.lo[a] = fmin(T.lo[a], c[a] - rad[k]);
            T.hi[a] = fmax(T.hi[a], c[a] + rad[k]);
        }
    }
    double ext[3], volume = 1;
    for (int a = 0; a < 3; ++a) {
        T.lo[a] -= margin; T.hi[a] += margin;
        ext[a] = T.hi[a] - T.lo[a];
        volume *= ext[a];
    }
    const double edge = cbrt(volume / fmin((double)n, (double)kGridMaxCells));
    for (int a = 0; a < 3; ++a) T.n[a] = (int)fmax(1.0, fmin(64.0, floor(ext[a] / edge + 0.5)));
    while ((long long)T.n[0] * T.n[1] * T.n[2] > kGridMaxCells) {  // rounding went up: shrink the longest axis
        int a = T.n[0] >= T.n[1] && T.n[0] >= T.n[2] ? 0 : T.n[1] >= T.n[2] ? 1 : 2;
        T.n[a]--;
    }
    for (int a = 0; a < 3; ++a) { T.cs[a] = ext[a] / T.n[a]; T.invCs[a] = T.n[a] / ext[a]; }
    const int cellsTotal = T.n[0] * T.n[1] * T.n[2];
    std::vector<std::vector<unsigned short>> lists(cellsTotal);
    for (int k = 0; k < n; ++k) {
        const double *c = &sph[4 * (size_t)k];
        int i0[3], i1[3];
        for (int a = 0; a < 3; ++a) {  // one cell more on either side than the arithmetic says
            i0[a] = std::max(0, (int)floor((c[a] - rad[k] - T.lo[a]) * T.invCs[a]) - 1);
            i1[a] = std::min(T.n[a] - 1, (int)floor((c[a] + rad[k] - T.lo[a]) * T.invCs[a]) + 1);
        }
        for (int z = i0[2]; z <= i1[2]; ++z)
            for (int y = i0[1]; y <= i1[1]; ++y)
                for (int x = i0[0]; x <= i1[0]; ++x) {
                    const int cell[3] = {x, y, z};
                    double dist2 = 0;  // from the centre to the cell's box
                    for (int a = 0; a < 3; ++a) {
                        const double bl = T.lo[a] + cell[a] * T.cs[a], bh = T.lo[a] + (cell[a] + 1) * T.cs[a];
                        const double gap = c[a] < bl ? bl - c[a] : c[a] > bh ? c[a] - bh : 0;
                        dist2 += gap * gap;
                    }
                    const double reach = rad[k] + 1e-6 * (1 + rad[k]);
                    if (dist2 <= reach * reach) lists[(z * T.n[1] + y) * T.n[0] + x].push_back((unsigned short)k);
                }
    }
    T.cells.assign(2 * (size_t)cellsTotal, 0);
    T.refs.clear();
    for (int c = 0; c < cellsTotal; ++c) {
        const std::vector<unsigned short> &l = lists[c];
        if (l.size() > 255 || T.refs.size() >= (1u << 24)) return false;
        T.cells[2 * c] = (unsigned)(T.refs.size() << 8) | (unsigned)l.size();
        T.cells[2 * c + 1] = (l.size() > 0 ? l[0] : 0u) | ((l.size() > 1 ? l[1] : 0u) << 16);
        if (l.size() > 2) T.refs.insert(T.refs.end(), l.begin() + 2, l.end());
    }
    T.refs.push_back(0);  // the walk reads two entries at a time, up to two past a cell's last
    T.refs.push_back(0);
    if (T.refs.size() & 1) T.refs.push_back(0);
    T.aos.assign(4 * (size_t)n, 0.0f);
    for (int k = 0; k < n; ++k) {
        const double *c = &sph[4 * (size_t)k];
        float rho;
        if (!screenRho(c, c[3], rho)) return false;
        T.aos[4 * k] = (float)c[0]; T.aos[4 * k + 1] = (float)c[1]; T.aos[4 * k + 2] = (float)c[2]; T.aos[4 * k + 3] = rho;
    }
    return true;
}

// The entries of a two-level table as plain {x, y, z, w} quadruples (entry e at 4 e), for code in which every lane
// reads an entry of its own: the tile-list build of closestHitLists.
inline void unpackEntries(const std::vector<float> &packed, size_t first, size_t count, std::vector<float> &out) {
    out.resize(4 * count);
    for (size_t i = 0; i < count; ++i) {
        const size_t e = first + i;
        const float *pair = packed.data() + 16 * (e / 4) + 8 * ((e % 4) >> 1);
        const int h = (int)(e & 1);
        out[4 * i] = pair[0 + h]; out[4 * i + 1] = pair[2 + h]; out[4 * i + 2] = pair[4 + h]; out[4 * i + 3] = pair[6 + h];
    }
}

// Occluder lists (closestHitLists): for one light and every sphere k, the spheres a shadow ray that STARTS ON sphere k
// towards the light can meet before the light.  Seen from the light, sphere j is a cone (u_j, alpha_j) as in
// packConeTable, widened by the same margin m_j (a line through the light that misses the widened cone has a
// negative double-precision discriminant against sphere j in the reference, with room to spare).  A shadow ray whose
// origin p lies on sphere k (p = s + t d as sphereIntersect forms it, shapes.c:67-68: | |p - c|^2 - r^2 | is the
// rounding error of the discriminant, below 2e-15 S^2 for coordinates below S, so p is within 1e-13 S^2 / r of the
// surface; then nudged 1e-6 along the ray, mainTracing.c:163-167) runs along the line through p and the light, and
// the direction from the light to p lies inside the cone of sphere k grown by grow_k = 1e-9 S + 3e-6 + 1e-13 S^2 / r_k
// (half-angle B_k).  So sphere j can be met only if
//   * the cones (u_j, A_j = alpha_j + m_j) and (u_k, B_k) overlap ON THE SAME SIDE of the light -- a sphere the line
//     meets on the far side of the light is met beyond the light, and the caller (lighting(), mainTracing.c:210-215)
//     treats "closest hit beyond the light" exactly like "no hit"; and
//   * sphere j is not wholly farther from the light than every point of sphere k (dist_j - r_j > dist_k + r_k +
//     slack): every point of the ray between p and the light is at most |p - L| from the light, and beyond the light
//     the ray moves away from such a sphere (its direction makes less than a right angle with u_j's opposite).
// A sphere with the light inside, on or near it (cone wider than kWide) is on every list of that light, and a sphere
// k whose own cone is that wide gets the list of ALL spheres, so hits at or next to the light's position are found
// with the reference's arithmetic as ever.  What is closest among the listed spheres is then either the reference's
// closest hit (if that lies before the light) or lies beyond the light whenever the reference's does.
// Shapes on the light's avoid list (mainTracing.c:134, :210) are left out here, once, instead of being compared
// per ray; so is sphere k itself, which the walk looks at first (closestHitLists).  Entries are sphere numbers
// (16 bits), read four at a time: every list starts at a multiple of four entries (a CHUNK of 8 bytes); list (k)
// starts at chunk index[k] & 0xfffff and has index[k] >> 20 entries, length kListAll standing for "every sphere, k
// included" (only for lights that avoid nothing).  Returns false when the scene does not qualify.
constexpr unsigned kListAll = 4095;
struct LightLists {
    std::vector<unsigned> index;        // per sphere: first chunk in pool | length << 20
    std::vector<unsigned short> pool;   // (a multiple of 4 entries)
};

inline bool packLightLists(const std::vector<double> &sph, const double light[3], double scale,
                           const unsigned char *avoidRow, LightLists &T) {
    const int n = (int)(sph.size() / 4);
    if (n > 65535) return false;
    const double eBound = 1e-13 * (4 * scale) * (4 * scale);
    const double kWide = 1.4;
    std::vector<double> dir(3 * (size_t)n), dist(n), rad(n), grow(n), cosA(n), sinA(n), cosB(n), sinB(n);
    std::vector<unsigned char> wideA(n), wideB(n);
    for (int k = 0; k < n; ++k) {
        const double *c = &sph[4 * (size_t)k];
        const double w[3] = {c[0] - light[0], c[1] - light[1], c[2] - light[2]};
        const double len = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        const double r = sqrt(c[3]);
        dist[k] = len;
        rad[k] = r;
        grow[k] = 1e-9 * scale + 3e-6 + 1e-13 * scale * scale / r;  // (r = 0: infinite, the sphere lists everything)
        wideA[k] = wideB[k] = 1;
        cosA[k] = cosB[k] = -1; sinA[k] = sinB[k] = 0;
        dir[3 * (size_t)k] = 1; dir[3 * (size_t)k + 1] = 0; dir[3 * (size_t)k + 2] = 0;
        if (!(len > 0) || !std::isfinite(len) || !std::isfinite(r)) continue;  // always listed, lists everything
        for (int a = 0; a < 3; ++a) dir[3 * (size_t)k + a] = w[a] / len;
        if (len > r * (1 + 1e-9) + 1e-9 * scale) {
            const double A = asin(r / len) + 1.01 * sqrt(eBound) / len + 1e-7;
            if (A < kWide) { wideA[k] = 0; cosA[k] = cos(A); sinA[k] = sin(A); }
        }
        if (len > (r + grow[k]) * (1 + 1e-9) + 1e-9 * scale) {
            const double B = asin((r + grow[k]) / len) + 1e-7;
            if (B < kWide) { wideB[k] = 0; cosB[k] = cos(B); sinB[k] = sin(B); }
        }
    }
    // the list of every sphere k on its own (ranges of k on different threads), then all of them into the pool in order
    std::vector<std::vector<unsigned short>> lists((size_t)n);
    std::vector<unsigned char> listsAll((size_t)n, 0);
    std::atomic<bool> tooLong(false);
    parallelRanges(n, 128, [&](int kBegin, int kEnd) {
        for (int k = kBegin; k < kEnd && !tooLong.load(std::memory_order_relaxed); ++k) {
            std::vector<unsigned short> &list = lists[(size_t)k];
            bool all = wideB[k] != 0 && !avoidRow;  // (with an avoid list the list of everything else is written out)
            const double *uk = &dir[3 * (size_t)k];
            const double reach = (dist[k] + rad[k] + grow[k]) * (1 + 1e-12);
            const bool everything = wideB[k] != 0;  // no bound on where its shadow rays go: every sphere is a candidate
            for (int j = 0; j < n && !all; ++j) {
                if (j == k || (avoidRow && avoidRow[j])) continue;
                if (!wideA[j] && !everything) {
                    const double *uj = &dir[3 * (size_t)j];
                    const double c = uj[0] * uk[0] + uj[1] * uk[1] + uj[2] * uk[2];
                    // cos(A_j + B_k); both angles are below kWide, so the sum is below pi and the cosine decreases
                    if (c < cosA[j] * cosB[k] - sinA[j] * sinB[k] - 1e-12) continue;
                    // wholly behind sphere k as seen from the light (and within a right angle of it, see above)
                    if (dist[j] - rad[j] > reach && c > 0.5 && cosA[j] > 0.9 && cosB[k] > 0.9) continue;
                }
                list.push_back((unsigned short)j);
                if (list.size() >= kListAll) {
                    if (avoidRow) tooLong.store(true, std::memory_order_relaxed);  // (too long to write out, and the list of all spheres would undo the avoid list)
                    all = true;
                }
            }
            listsAll[(size_t)k] = all ? 1 : 0;
        }
    });
    if (tooLong.load()) return false;
    T.index.assign(n, 0);
    T.pool.clear();
    for (int k = 0; k < n; ++k) {
        const std::vector<unsigned short> &list = lists[(size_t)k];
        if (listsAll[(size_t)k]) {
            T.index[k] = kListAll << 20;
        } else {
            if ((T.pool.size() + list.size()) / 4 + 1 >= (1u << 20)) return false;
            T.index[k] = (unsigned)(T.pool.size() / 4) | ((unsigned)list.size() << 20);
            T.pool.insert(T.pool.end(), list.begin(), list.end());
            T.pool.resize((T.pool.size() + 3) / 4 * 4, 0);
        }
    }
    return true;
}

// The camera table of an ORTHOGRAPHIC view (closestHitLists): every primary ray has the same direction d, so what a ray
// can meet depends on where its line crosses the plane through the screen centre perpendicular to d.  With P = I - d d^T
// sphere j is the point q_j = P (c_j - screenCenter) of that plane and a line crossing it at w meets the sphere iff
// |q_j - w| <= r_j.  An entry is {q, R} (single precision, R rounded up); group entries are bounding balls of their
// members' (k-d leaves of the q).  R carries r_j, the margin m = 1.01 sqrt(eBound) (a sphere left out is missed by more
// than m, so its double-precision discriminant in the reference, r^2 - distance^2 up to rounding below eBound, is
// negative) and the rounding of q to float.  Entries as plain quadruples: groups[4 g], members[4 (g kGroup + i)],
// memberShape -1 for padding (R = -inf: never passes).
struct OrthoTable {
    std::vector<float> groups, members;
    std::vector<int> memberShape;
    int kGroup = 0;
};

inline void packOrthoTable(const std::vector<double> &sph, const double centre[3], const double dir[3], double scale,
                           int kGroup, OrthoTable &T) {
    const int n = (int)(sph.size() / 4);
    const double eBound = 1e-13 * (4 * scale) * (4 * scale);
    const double margin = 1.01 * sqrt(eBound) + 1e-9 * scale;
    const double roundOff = 8 * 5.9604644775390625e-08 * scale;  // (three coordinates of up to 2 * scale rounded to float, and R itself)
    std::vector<double> q(3 * (size_t)n), R(n);
    for (int k = 0; k < n; ++k) {
        const double *c = &sph[4 * (size_t)k];
        const double v[3] = {c[0] - centre[0], c[1] - centre[1], c[2] - centre[2]};
        const double along = v[0] * dir[0] + v[1] * dir[1] + v[2] * dir[2];
        for (int a = 0; a < 3; ++a) q[3 * (size_t)k + a] = v[a] - along * dir[a];
        R[k] = sqrt(c[3]) + margin + roundOff;
    }
    const std::vector<int> order = kdLeafOrder(q, n, kGroup);
    const int nGroups = (n + kGroup - 1) / kGroup;
    T.kGroup = kGroup;
    T.groups.assign(4 * (size_t)nGroups, 0.0f);
    T.members.assign(4 * (size_t)nGroups * kGroup, 0.0f);
    T.memberShape.assign((size_t)nGroups * kGroup, -1);
    for (size_t i = 0; i < (size_t)nGroups * kGroup; ++i) T.members[4 * i + 3] = -INFINITY;
    auto up = [](double x) { float f = (float)x; return (double)f < x ? nextafterf(f, INFINITY) : f; };
    for (int gI = 0; gI < nGroups; ++gI) {
        const int first = gI * kGroup, last = std::min(n, first + kGroup);
        double g[3] = {0, 0, 0};
        for (int m = first; m < last; ++m)
            for (int a = 0; a < 3; ++a) g[a] += q[3 * (size_t)order[m] + a] / (last - first);
        double radius = 0;
        for (int m = first; m < last; ++m) {
            const int k = order[m];
            const double *p = &q[3 * (size_t)k];
            radius = fmax(radius, sqrt((p[0] - g[0]) * (p[0] - g[0]) + (p[1] - g[1]) * (p[1] - g[1]) + (p[2] - g[2]) * (p[2] - g[2])) + R[k]);
            float *e = &T.members[4 * ((size_t)gI * kGroup + (m - first))];
            e[0] = (float)p[0]; e[1] = (float)p[1]; e[2] = (float)p[2]; e[3] = up(R[k]);
            T.memberShape[(size_t)gI * kGroup + (m - first)] = k;
        }
        float *e = &T.groups[4 * (size_t)gI];
        e[0] = (float)g[0]; e[1] = (float)g[1]; e[2] = (float)g[2]; e[3] = up(radius * (1 + 1e-12) + roundOff);
    }
}

// Interleave the tables of several ray families block by block (a block = 4 packed entries = 64 bytes): block b of
// family f lands at float4 index 4 * (b * families + f), and the shape index of member entry i at i * families + f.
// A lane then reads a block with one address and four fixed offsets, and lanes of one wave that trace rays of
// different families read different 64-byte slots.
inline void interleaveFamilies(const std::vector<const TwoLevelTable *> &fam, std::vector<float> &entries,
                               std::vector<int> &memberShape) {
    const size_t F = fam.size(), total = fam[0]->entries.size() / 4, members = fam[0]->memberShape.size();
    entries.assign(4 * total * F, 0.0f);
    memberShape.assign(members * F, -1);
    for (size_t f = 0; f < F; ++f) {
        for (size_t e = 0; e < total; ++e)
            for (int c = 0; c < 4; ++c) entries[4 * (((e / 4) * F + f) * 4 + e % 4) + c] = fam[f]->entries[4 * e + c];
        for (size_t i = 0; i < members; ++i) memberShape[i * F + f] = fam[f]->memberShape[i];
    }
}

// The same interleaving for the third-level sections.
inline void interleaveSupers(const std::vector<const TwoLevelTable *> &fam, std::vector<float> &entries) {
    const size_t F = fam.size(), total = fam[0]->supers.size() / 4;
    entries.assign(4 * total * F, 0.0f);
    for (size_t f = 0; f < F; ++f)
        for (size_t e = 0; e < total; ++e)
            for (int c = 0; c < 4; ++c) entries[4 * (((e / 4) * F + f) * 4 + e % 4) + c] = fam[f]->supers[4 * e + c];
}

}  // namespace graft
