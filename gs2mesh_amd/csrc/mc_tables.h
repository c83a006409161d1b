// mc_tables.h -- marching-cubes case table (host code; copied to the device at create).
//
// Corner / edge numbering is the usual one and Open3D's (MarchingCubesConst.h `shift` / `edge_to_vert`): corners
// 0:(0,0,0) 1:(1,0,0) 2:(1,1,0) 3:(0,1,0) 4:(0,0,1) 5:(1,0,1) 6:(1,1,1) 7:(0,1,1); edges 0:0-1 1:1-2 2:2-3 3:3-0 4:4-5
// 5:5-6 6:6-7 7:7-4 8:0-4 9:1-5 10:2-6 11:3-7.  A corner is "inside" when tsdf < 0 (bit set in the case index).
//
// The triangulation of every case is the CLASSIC table (Lorensen & Cline's cases as tabulated by Bloyd / Bourke, public
// domain) -- the `tri_table` Open3D 0.17 ships -- so that the mesh has Open3D's triangles, not only its vertices.  The table
// (mc_classic.inc) is produced by tools/mc_classic_table.py, which first verifies it: every row uses exactly the cut edges
// of its case, its patches close into loops on the cube's faces, and all 820 triangles are wound alike.  Like Open3D
// (ScalableTSDFVolume::ExtractTriangleMesh) a triangle is emitted as (e[i], e[i+2], e[i+1]): its normal points from the
// inside (tsdf < 0) to the outside.  (Round 1 derived a hole-free table of its own; it triangulated the same loops with
// other diagonals and resolved ambiguous faces differently, i.e. not Open3D's mesh.)
#pragma once
#include <string.h>

#define GS2M_MC_MAX_TRIS 5  // the classic table has at most 5 triangles per cube

struct McTables {
    signed char tri[256][GS2M_MC_MAX_TRIS * 3];  // edge ids, 3 per triangle, in emission order
    unsigned char ntri[256];
};

static const int mc_corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
// edge -> (lower corner, upper corner along its axis, axis)
static const int mc_edge[12][3] = {{0, 1, 0}, {1, 2, 1}, {3, 2, 0}, {0, 3, 1}, {4, 5, 0}, {5, 6, 1},
                                   {7, 6, 0}, {4, 7, 1}, {0, 4, 2}, {1, 5, 2}, {2, 6, 2}, {3, 7, 2}};

static const signed char mc_classic_tri_table[256][16] = {
#include "mc_classic.inc"
};

static inline bool mc_generate(McTables* T) {
    memset(T, 0, sizeof(*T));
    for (int c = 0; c < 256; ++c) {
        int nt = 0;
        for (int i = 0; i + 2 < 16 && mc_classic_tri_table[c][i] >= 0; i += 3) {
            if (nt >= GS2M_MC_MAX_TRIS) return false;
            // Open3D pushes (edge_to_index[t[i]], edge_to_index[t[i + 2]], edge_to_index[t[i + 1]])
            T->tri[c][3 * nt + 0] = mc_classic_tri_table[c][i];
            T->tri[c][3 * nt + 1] = mc_classic_tri_table[c][i + 2];
            T->tri[c][3 * nt + 2] = mc_classic_tri_table[c][i + 1];
            nt++;
        }
        T->ntri[c] = (unsigned char)nt;
    }
    return T->ntri[0] == 0 && T->ntri[255] == 0;
}
