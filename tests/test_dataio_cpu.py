"""include/dataio.hpp (drop-in for the reference's DataIo<PointT>, include/dataio.hpp:26-627): PCD / PLY / TXT files in the
layouts PCL writes are produced here by an independent writer, read by the C++ header, written back and re-parsed."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_pcd_binary_pcl_layout(path, xyz, inten):
    """What pcl::io::savePCDFileBinary writes for PointXYZI: the 32-byte struct with '_' padding fields."""
    n = len(xyz)
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z _ intensity _\nSIZE 4 4 4 1 4 1\nTYPE F F F U F U\n"
           "COUNT 1 1 1 4 1 12\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n" % (n, n))
    rec = np.zeros((n, 8), np.float32)
    rec[:, :3] = xyz
    rec[:, 4] = inten
    with open(path, "wb") as f:
        f.write(hdr.encode())
        f.write(rec.tobytes())


def write_pcd_ascii(path, xyz, inten):
    n = len(xyz)
    with open(path, "w") as f:
        f.write("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
                "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA ascii\n" % (n, n))
        for p, i in zip(xyz, inten):
            f.write("%.9g %.9g %.9g %.9g\n" % (p[0], p[1], p[2], i))


def write_ply(path, xyz, inten, binary):
    n = len(xyz)
    hdr = ("ply\nformat %s 1.0\ncomment test\nelement vertex %d\nproperty double x\nproperty double y\nproperty double z\nproperty uchar red\n"
           "property float intensity\nelement camera 1\nproperty float view_px\nend_header\n" % ("binary_little_endian" if binary else "ascii", n))
    with open(path, "wb") as f:
        f.write(hdr.encode())
        for p, i in zip(xyz, inten):
            if binary:
                f.write(struct.pack("<dddBf", float(p[0]), float(p[1]), float(p[2]), 7, float(i)))
            else:
                f.write(("%.17g %.17g %.17g 7 %.9g\n" % (p[0], p[1], p[2], i)).encode())
        f.write(struct.pack("<f", 0.0) if binary else b"0\n")


def parse_pcd(path):
    raw = open(path, "rb").read()
    head, body = raw.split(b"DATA binary\n", 1)
    fields = [l for l in head.decode().splitlines() if l.startswith("FIELDS")][0].split()[1:]
    n = int([l for l in head.decode().splitlines() if l.startswith("POINTS")][0].split()[1])
    return fields, np.frombuffer(body, np.float32).reshape(n, len(fields))


def parse_ply_ascii(path):
    lines = open(path).read().splitlines()
    e = lines.index("end_header")
    n = int([l for l in lines[:e] if l.startswith("element vertex")][0].split()[2])
    return np.array([[float(v) for v in l.split()] for l in lines[e + 1:e + 1 + n]])


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("dataio") / "test_dataio"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_dataio.cpp"),
                           "-o", str(out)])
    return str(out)


def test_dataio_formats_round_trip(exe, tmp_path):
    rng = np.random.default_rng(5)
    xyz = (rng.normal(0, 30, (257, 3))).astype(np.float32)
    inten = rng.uniform(0, 1, 257).astype(np.float32)
    files = [tmp_path / n for n in ("a.pcd", "b.pcd", "c.ply", "d.ply", "e.txt")]
    write_pcd_binary_pcl_layout(files[0], xyz, inten)
    write_pcd_ascii(files[1], xyz, inten)
    write_ply(files[2], xyz, inten, binary=False)
    write_ply(files[3], xyz, inten, binary=True)
    np.savetxt(files[4], xyz.astype(np.float64), fmt="%.6f")
    out = subprocess.run([exe, str(tmp_path)] + [str(f) for f in files], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    reads = [l.split() for l in out.stdout.splitlines() if l.startswith("READ")]
    assert len(reads) == 5
    sums = xyz.astype(np.float64).sum(0)
    for r in reads:
        assert r[2] == "1" and int(r[3]) == 257
        tol = 1e-3 if r[1].endswith(".txt") else 1e-6  # the txt file carries 6 decimals
        np.testing.assert_allclose([float(v) for v in r[4:7]], sums, atol=257 * tol)
        if not r[1].endswith(".txt"):
            np.testing.assert_allclose(float(r[7]), inten.astype(np.float64).sum(), rtol=1e-6)
    # what the header wrote
    fields, rec = parse_pcd(tmp_path / "out.pcd")
    assert fields == ["x", "y", "z", "intensity"]
    np.testing.assert_array_equal(rec[:, :3], xyz)
    np.testing.assert_array_equal(rec[:, 3], inten)
    ply = parse_ply_ascii(tmp_path / "out.ply")
    np.testing.assert_array_equal(ply[:, :3].astype(np.float32), xyz)  # 9 significant digits round-trip a float
    np.testing.assert_array_equal(ply[:, 3].astype(np.float32), inten)
    txt = np.loadtxt(tmp_path / "out.txt")
    np.testing.assert_allclose(txt, xyz, atol=5.1e-7 + 1e-6 * 0)  # setprecision(6), fixed (dataio.hpp:544-546)
    assert open(tmp_path / "out.txt").readline().count("  ") == 2  # two spaces between the columns, as the reference writes
    sub = np.loadtxt(tmp_path / "out_sub.txt")
    np.testing.assert_allclose(sub, xyz[::3], atol=5.1e-7)
    kp = np.loadtxt(tmp_path / "kp.txt")
    np.testing.assert_allclose(kp, xyz[[2, 0, 5]], atol=5.1e-7)
    coord = [l.split() for l in out.stdout.splitlines() if l.startswith("COORD")][0]
    assert int(coord[1]) == 3
    np.testing.assert_allclose([float(v) for v in coord[2:]], [xyz[2, 0], xyz[0, 1], xyz[5, 2]], rtol=1e-7)
    assert "UNDEF 0" in out.stdout and "Undefined Point Cloud Format." in out.stdout
