"""Writes tests/golden/boatlike.obj: a procedural boat-shaped mesh with the attribute set of the reference's
resources/example-3d-models/boat/boat.obj (Blender export: v / vt / vn, `v/vt/vn` face corners, quads and n-gons that the
loader must fan-triangulate, `o` / `s` / `usemtl` lines to skip) and about the same size (~800 triangles, 4 units long, keel
at y = 0.08, gunwale at y = 0.5).  It is DATA for tests of `Mesh.Load` + the rasterizer; no reference file is copied.

    python tools/make_boatlike_obj.py
"""
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NS, NR = 28, 12                     # stations along the hull, points per half rib


def half_width(t):                  # t in [0,1] bow..stern
    return 0.46 * math.sin(math.pi * min(max(t * 0.92 + 0.04, 0.0), 1.0)) ** 0.65


def main():
    V, VT, VN, F = [], [], [], []

    def v(p):
        V.append(p)
        return len(V)

    def vt(p):
        VT.append(p)
        return len(VT)

    def vn(p):
        l = math.sqrt(sum(c * c for c in p)) or 1.0
        VN.append(tuple(c / l for c in p))
        return len(VN)
    # hull surface: ribs of a rounded V section, both sides
    grid = {}
    for i in range(NS + 1):
        t = i / NS
        z = -2.05 + 4.1 * t
        hw = half_width(t)
        sheer = 0.44 + 0.10 * (2 * t - 1) ** 2
        for side in (-1, 1):
            for j in range(NR + 1):
                a = j / NR                                  # 0 keel .. 1 gunwale
                x = side * hw * (a ** 0.7)
                y = 0.085 + (sheer - 0.085) * (a ** 1.8)
                nx, ny = side * (1.8 * a ** 0.8 + 0.05), -(1.0 - a) - 0.15
                grid[(i, side, j)] = (v((x, y, z)), vt((0.5 + side * 0.45 * a, t)), vn((nx, ny, 0.15 * (2 * t - 1))))
    for i in range(NS):
        for side in (-1, 1):
            for j in range(NR):
                q = [grid[(i, side, j)], grid[(i + 1, side, j)], grid[(i + 1, side, j + 1)], grid[(i, side, j + 1)]]
                if side == 1:
                    q = q[::-1]
                if (i + j) % 5 == 0:                       # some quads come pre-split, as a mixed export has them
                    F.append([q[0], q[1], q[2]])
                    F.append([q[0], q[2], q[3]])
                else:
                    F.append(q)
    # deck: one n-gon strip per pair of stations + thwarts (boxes)
    up = vn((0, 1, 0))
    for i in range(2, NS - 2, 2):
        ring = []
        for ii, side in ((i, -1), (i + 1, -1), (i + 2, -1), (i + 2, 1), (i + 1, 1), (i, 1)):
            vi, ti, _ = grid[(ii, side, NR - 2)]
            ring.append((vi, ti, up))
        F.append(ring)                                      # hexagon
    for zc in (-0.9, 0.1, 1.0):                             # three thwarts
        hw = half_width((zc + 2.05) / 4.1) * 0.8
        c = [(-hw, 0.36, zc - 0.08), (hw, 0.36, zc - 0.08), (hw, 0.36, zc + 0.08), (-hw, 0.36, zc + 0.08),
             (-hw, 0.40, zc - 0.08), (hw, 0.40, zc - 0.08), (hw, 0.40, zc + 0.08), (-hw, 0.40, zc + 0.08)]
        ids = [v(p) for p in c]
        tx = [vt((0.1 + 0.8 * (k % 4 in (1, 2)), 0.1 + 0.05 * (k // 4) + 0.8 * (k % 4 in (2, 3)) * 0.1)) for k in range(8)]
        for quad, n in (((4, 5, 6, 7), (0, 1, 0)), ((0, 3, 2, 1), (0, -1, 0)), ((0, 1, 5, 4), (0, 0, -1)), ((2, 3, 7, 6), (0, 0, 1)),
                        ((1, 2, 6, 5), (1, 0, 0)), ((3, 0, 4, 7), (-1, 0, 0))):
            ni = vn(n)
            F.append([(ids[k], tx[k], ni) for k in quad][::-1])
    # mast: an octagonal prism closed by an 8-gon, written with NEGATIVE (relative) indices
    base = len(V)
    rim_t = vt((0.5, 0.5))
    for k in range(8):
        a = 2 * math.pi * k / 8
        v((0.03 * math.cos(a), 0.40, -0.3 + 0.03 * math.sin(a)))
        v((0.03 * math.cos(a), 1.30, -0.3 + 0.03 * math.sin(a)))
    rel = []
    for k in range(8):
        a = 2 * math.pi * (k + 0.5) / 8
        ni = vn((math.cos(a), 0, math.sin(a)))
        b0, t0, b1, t1 = 2 * k, 2 * k + 1, 2 * ((k + 1) % 8), 2 * ((k + 1) % 8) + 1
        rel.append(("quad", [(b0, ni), (t0, ni), (t1, ni), (b1, ni)]))
    out = ["# boat-like test mesh (procedural; tools/make_boatlike_obj.py)", "o boatlike_hull"]
    out += ["v %.6f %.6f %.6f" % p for p in V]
    out += ["vt %.6f %.6f" % p for p in VT]
    out += ["vn %.4f %.4f %.4f" % p for p in VN]
    out += ["usemtl hull", "s 1"]
    for f in F:
        out.append("f " + " ".join("%d/%d/%d" % c for c in f))
    out += ["o boatlike_mast", "s off"]
    nV = len(V)
    for _, corners in rel:
        out.append("f " + " ".join("%d/%d/%d" % (base + c + 1 - nV - 1, rim_t, ni) for c, ni in corners))
    cap = vn((0, 1, 0))
    out.append("f " + " ".join("%d/%d/%d" % (base + 2 * k + 1 + 1 - nV - 1, rim_t, cap) for k in reversed(range(8))))
    # the cap's normal was appended after the vn block was emitted: add it now (OBJ allows interleaving)
    out.insert(out.index("usemtl hull"), "vn %.4f %.4f %.4f" % VN[-1])
    dst = os.path.join(ROOT, "tests", "golden", "boatlike.obj")
    with open(dst, "w") as fh:
        fh.write("\n".join(out) + "\n")
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB;", len(V), "v,", len(F) + 9, "faces")


if __name__ == "__main__":
    main()
