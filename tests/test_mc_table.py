"""The marching-cubes case table shipped in gs2mesh_amd/csrc/mc_classic.inc is the classic 256-case table (the one Open3D
0.17 holds in MarchingCubesConst.h); tools/mc_classic_table.py verifies it structurally -- exact cut-edge sets, closed loops on
the cube faces, uniform winding -- and this test re-runs those checks and that the .inc is the verified table."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_classic_table_is_consistent_and_is_what_ships():
    import mc_classic_table as m
    listed_outward = m.check()
    assert listed_outward is False        # as listed the normals point into the solid; Open3D (and we) emit (i, i+2, i+1)
    assert sum(len(r) // 3 for r in m.T) == 820
    rows = []
    for line in open(os.path.join(ROOT, "gs2mesh_amd", "csrc", "mc_classic.inc")):
        line = line.strip()
        if line.startswith("{"):
            rows.append([int(v) for v in line.strip("{},").split(",")])
    assert len(rows) == 256
    for c, (row, ref) in enumerate(zip(rows, m.T)):
        assert row == ref + [-1] * (16 - len(ref)), c
