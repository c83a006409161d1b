// mc_tables.h -- marching-cubes case table, GENERATED at library initialisation (host code).
//
// Corner / edge numbering is the usual one (and Open3D's): corners 0:(0,0,0) 1:(1,0,0) 2:(1,1,0)
// 3:(0,1,0) 4:(0,0,1) 5:(1,0,1) 6:(1,1,1) 7:(0,1,1); edges 0:0-1 1:1-2 2:2-3 3:3-0 4:4-5 5:5-6 6:6-7 7:7-4
// 8:0-4 9:1-5 10:2-6 11:3-7.  A corner is "inside" when tsdf < 0 (bit set in the case index).
//
// Instead of shipping a hand-made 256 x 16 table, the triangulation of every case is derived:
//   1. on each of the 6 faces the intersected edges are joined by segments; a face with 4
//      intersected edges (diagonal corners inside) is resolved by cutting off each INSIDE corner
//      separately -- the rule only looks at the face's own corner signs, so two cubes sharing a face
//      always agree and the surface has no holes (the classic table is not consistent there);
//   2. the segments chain into closed loops (every intersected edge belongs to exactly two faces);
//   3. each loop is oriented so that its normal points from the inside corners to the outside
//      (towards positive tsdf) and fan-triangulated.
// Vertex positions (on cube edges) are the same as with any marching-cubes table; only the choice of
// diagonals inside a loop and the ambiguous-face resolution can differ from Open3D's table.
#pragma once
#include <string.h>

#define GS2M_MC_MAX_TRIS 8  // generated maximum is checked at init

struct McTables {
    signed char tri[256][GS2M_MC_MAX_TRIS * 3];  // edge ids, 3 per triangle
    unsigned char ntri[256];
};

static const int mc_corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
// edge -> (lower corner, upper corner along its axis, axis)
static const int mc_edge[12][3] = {{0, 1, 0}, {1, 2, 1}, {3, 2, 0}, {0, 3, 1}, {4, 5, 0}, {5, 6, 1},
                                   {7, 6, 0}, {4, 7, 1}, {0, 4, 2}, {1, 5, 2}, {2, 6, 2}, {3, 7, 2}};
// faces: 4 corners in cyclic order and the 4 edges between consecutive corners
static const int mc_face_corner[6][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 1, 5, 4}, {3, 2, 6, 7}, {0, 3, 7, 4}, {1, 2, 6, 5}};
static const int mc_face_edge[6][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 9, 4, 8}, {2, 10, 6, 11}, {3, 11, 7, 8}, {1, 10, 5, 9}};

static inline bool mc_generate(McTables* T) {
    memset(T, 0, sizeof(*T));
    for (int c = 0; c < 256; ++c) {
        // segment endpoints per edge (each intersected edge gets exactly two)
        int link[12][2];
        int nlink[12];
        for (int e = 0; e < 12; ++e) nlink[e] = 0, link[e][0] = link[e][1] = -1;
        auto add = [&](int a, int b) {
            link[a][nlink[a]++] = b;
            link[b][nlink[b]++] = a;
        };
        for (int f = 0; f < 6; ++f) {
            int in[4], cut[4], ncut = 0;
            for (int k = 0; k < 4; ++k) in[k] = (c >> mc_face_corner[f][k]) & 1;
            for (int k = 0; k < 4; ++k) cut[k] = in[k] != in[(k + 1) & 3], ncut += cut[k];
            if (ncut == 2) {
                int a = -1, b = -1;
                for (int k = 0; k < 4; ++k)
                    if (cut[k]) (a < 0 ? a : b) = mc_face_edge[f][k];
                add(a, b);
            } else if (ncut == 4) {
                // corners alternate; cut off each inside corner k: its two incident face edges are k-1 and k
                for (int k = 0; k < 4; ++k)
                    if (in[k]) add(mc_face_edge[f][(k + 3) & 3], mc_face_edge[f][k]);
            }
        }
        bool used[12] = {false};
        int nt = 0;
        for (int e0 = 0; e0 < 12; ++e0) {
            if (nlink[e0] == 0 || used[e0]) continue;
            if (nlink[e0] != 2) return false;
            int loop[12], n = 0, prev = -1, cur = e0;
            while (true) {
                loop[n++] = cur;
                used[cur] = true;
                int nxt = link[cur][0] != prev ? link[cur][0] : link[cur][1];
                if (n > 1 && link[cur][0] == link[cur][1]) nxt = link[cur][0];  // 2-cycle guard (cannot occur)
                prev = cur;
                cur = nxt;
                if (cur == e0) break;
                if (n >= 12) return false;
            }
            if (n < 3) return false;
            // orientation: Newell normal of the loop of edge midpoints vs the direction to an inside corner
            double P[12][3], m[3] = {0, 0, 0}, nrm[3] = {0, 0, 0};
            for (int k = 0; k < n; ++k)
                for (int a = 0; a < 3; ++a) {
                    P[k][a] = 0.5 * (mc_corner[mc_edge[loop[k]][0]][a] + mc_corner[mc_edge[loop[k]][1]][a]);
                    m[a] += P[k][a] / n;
                }
            for (int k = 0; k < n; ++k) {
                const double* p = P[k];
                const double* q = P[(k + 1) % n];
                nrm[0] += (p[1] - q[1]) * (p[2] + q[2]);
                nrm[1] += (p[2] - q[2]) * (p[0] + q[0]);
                nrm[2] += (p[0] - q[0]) * (p[1] + q[1]);
            }
            // inside endpoint of the loop's first edge
            const int lo = mc_edge[loop[0]][0], hi = mc_edge[loop[0]][1];
            const int vin = ((c >> lo) & 1) ? lo : hi;
            double d = 0;
            for (int a = 0; a < 3; ++a) d += nrm[a] * (mc_corner[vin][a] - m[a]);
            if (d > 0)  // normal points at the inside corner: reverse
                for (int k = 0; k < n / 2; ++k) {
                    int t = loop[k];
                    loop[k] = loop[n - 1 - k];
                    loop[n - 1 - k] = t;
                }
            for (int k = 1; k + 1 < n; ++k) {
                if (nt >= GS2M_MC_MAX_TRIS) return false;
                T->tri[c][3 * nt + 0] = (signed char)loop[0];
                T->tri[c][3 * nt + 1] = (signed char)loop[k];
                T->tri[c][3 * nt + 2] = (signed char)loop[k + 1];
                nt++;
            }
        }
        T->ntri[c] = (unsigned char)nt;
    }
    return true;
}
