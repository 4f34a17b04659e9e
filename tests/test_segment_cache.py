"""Segment cache files of Line3D::addImage (line3D.cc:143-168): the product's boost-free reader / writer
(C ABI, l3d_segcache.cpp) against the oracle's independent pure-Python restatement of the archive layout, a hand-assembled
known-answer file, and the failure cases.  The layout itself is unpinned by the reference (no boost here, no sample file)."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import l3d_oracle_sfm as osfm  # noqa: E402
from line3d_amd import io as lio  # noqa: E402


def _scene(n, seed, density=0.2):
    rng = np.random.default_rng(seed)
    segs = rng.uniform(0, 1000, (n, 4)).astype(np.float32)
    coll = {}
    for i in range(n):
        for j in range(i + 1, n):
            if rng.uniform() < density:
                w = np.float32(rng.uniform(0.01, 1.0))
                coll.setdefault(i, {})[j] = w
                coll.setdefault(j, {})[i] = w                          # segments.h:91-92: both directions
    return segs, coll


def _directed(coll):
    e = [(i, j, coll[i][j]) for i in sorted(coll) for j in sorted(coll[i])]
    return (np.array([x[0] for x in e], np.int32), np.array([x[1] for x in e], np.int32), np.array([x[2] for x in e], np.float32))


def test_file_name_follows_addImage():
    assert lio.segment_cache_filename(17, 1920, 1080, True) == "/segments_17_1920x1080_coll1.bin" == osfm.filename_segment_cache(17, 1920, 1080, True)
    assert lio.segment_cache_filename(3, 640, 480, False) == "/segments_3_640x480_coll0.bin" == osfm.filename_segment_cache(3, 640, 480, False)


@pytest.mark.parametrize("n,density,lib", [(1, 0.0, 9), (7, 0.0, 12), (40, 0.2, 12), (300, 0.02, 17), (65, 1.0, 19)])
def test_reader_and_writer_agree_with_the_oracle(tmp_path, n, density, lib):
    segs, coll = _scene(n, 100 + n, density)
    ci, cj, cw = _directed(coll)
    a, b = str(tmp_path / "oracle.bin"), str(tmp_path / "product.bin")
    osfm.write_segment_cache(a, segs, coll, library_version=lib)
    # entries handed to the writer in scrambled order: the file holds them in std::map order
    perm = np.random.default_rng(1).permutation(len(ci))
    lio.write_segment_cache(b, segs, ci[perm], cj[perm], cw[perm], library_version=lib)
    assert open(a, "rb").read() == open(b, "rb").read()
    got = lio.read_segment_cache(a)
    assert got.library_version == lib
    assert np.array_equal(got.segments, segs)
    assert np.array_equal(got.ci, ci) and np.array_equal(got.cj, cj) and np.array_equal(got.cw, cw)
    osegs, ocoll, olib = osfm.read_segment_cache(b)
    assert olib == lib and np.array_equal(osegs, segs)
    assert {i: dict(r) for i, r in ocoll.items()} == {i: dict(r) for i, r in coll.items()}


def test_known_answer_bytes(tmp_path):
    """Two segments that are collinear with each other, assembled by hand from the documented layout."""
    pre = b"\x00" + b"\x00\x00\x00\x00"
    blob = (struct.pack("<Q", 22) + b"serialization::archive" + b"\x0c\x00" + b"\x04\x08\x04\x08" + b"\x01\x00\x00\x00"
            + pre                                                       # L3DSegments
            + pre + struct.pack("<Q", 2) + struct.pack("<I", 0)         # outer map: 2 items, item version 0
            + pre + struct.pack("<I", 0)                                # first item: pair preamble, key 0
            + pre + struct.pack("<Q", 1) + struct.pack("<I", 0)         #   inner map: 1 item
            + pre + struct.pack("<I", 1) + struct.pack("<f", 0.5)       #   pair preamble, (1, 0.5)
            + struct.pack("<I", 1) + struct.pack("<Q", 1) + struct.pack("<I", 0) + struct.pack("<I", 0) + struct.pack("<f", 0.5)   # key 1: {0: 0.5}
            + b"\x05\x00" + b"\x01" + b"\x00\x00\x00\x00" + b"\x00\x00\x00\x00"       # pointer: class id 5, tracked, version 0, object id 0
            + struct.pack("<III", 4, 2, 8) + struct.pack("<QQQQ", 32, 8, 0, 0)
            + struct.pack("<8f", 1, 2, 3, 4, 0, 0, 0, 0) + struct.pack("<8f", 5, 6, 7, 8, 0, 0, 0, 0))
    p = str(tmp_path / "kat.bin")
    open(p, "wb").write(blob)
    got = lio.read_segment_cache(p)
    assert got.library_version == 12
    assert got.segments.tolist() == [[1, 2, 3, 4], [5, 6, 7, 8]]
    assert got.ci.tolist() == [0, 1] and got.cj.tolist() == [1, 0] and got.cw.tolist() == [0.5, 0.5]
    q = str(tmp_path / "kat_written.bin")
    lio.write_segment_cache(q, got.segments, got.ci, got.cj, got.cw, library_version=12)
    assert open(q, "rb").read() == blob
    osegs, ocoll, _ = osfm.read_segment_cache(p)
    assert osegs.tolist() == got.segments.tolist() and ocoll == {0: {1: np.float32(0.5)}, 1: {0: np.float32(0.5)}}


def test_default_constructed_segments_have_a_null_pointer(tmp_path):
    """L3DSegments() (segments.h:62-64) serialises segments_ == NULL as class id -1."""
    pre = b"\x00\x00\x00\x00\x00"
    blob = (struct.pack("<Q", 22) + b"serialization::archive" + b"\x0a\x00\x04\x08\x04\x08\x01\x00\x00\x00" + pre + pre
            + struct.pack("<QI", 0, 0) + b"\xff\xff")
    p = str(tmp_path / "null.bin")
    open(p, "wb").write(blob)
    got = lio.read_segment_cache(p)
    assert got.segments.shape == (0, 4) and len(got.ci) == 0 and got.library_version == 10


def test_malformed_files_are_refused_with_a_message(tmp_path):
    segs, coll = _scene(12, 5, 0.3)
    good = str(tmp_path / "good.bin")
    osfm.write_segment_cache(good, segs, coll)
    data = open(good, "rb").read()
    cases = {
        "missing": None,
        "empty": b"",
        "signature": data[:8] + b"Serialization::archive" + data[30:],
        "old library": data[:30] + b"\x05\x00" + data[32:],
        "foreign sizes": data[:32] + b"\x04\x04\x04\x08" + data[36:],
        "big endian": data[:36] + b"\x00\x00\x00\x01" + data[40:],
        "truncated map": data[:80],
        "truncated rows": data[:-4],
        "trailing bytes": data + b"\x00",
        "class version": data[:41] + b"\x01" + data[42:],
    }
    for name, blob in cases.items():
        p = str(tmp_path / (name.replace(" ", "_") + ".bin"))
        if blob is not None:
            open(p, "wb").write(blob)
        with pytest.raises(RuntimeError) as e:
            lio.read_segment_cache(p)
        assert str(e.value), name
    # a collinearity entry that names a segment the array does not hold
    bad = {0: {50: np.float32(0.5)}, 50: {0: np.float32(0.5)}}
    p = str(tmp_path / "dangling.bin")
    osfm.write_segment_cache(p, segs, bad)
    with pytest.raises(RuntimeError):
        lio.read_segment_cache(p)
    # the writer refuses what a std::map cannot hold / what the array cannot index
    with pytest.raises(RuntimeError):
        lio.write_segment_cache(str(tmp_path / "dup.bin"), segs, [0, 0], [1, 1], [0.5, 0.5])
    with pytest.raises(RuntimeError):
        lio.write_segment_cache(str(tmp_path / "range.bin"), segs, [0], [12], [0.5])


def test_reader_survives_random_corruption(tmp_path):
    """Byte flips, truncations and spliced garbage: the reader either parses the file or refuses it with a message -- it never
    crashes and never trusts a count or a size the file cannot back (e.g. 2^31 rows of 2^31 floats, whose byte count wraps)."""
    segs, coll = _scene(30, 9, 0.25)
    good = str(tmp_path / "good.bin")
    osfm.write_segment_cache(good, segs, coll)
    data = bytearray(open(good, "rb").read())
    rng = np.random.default_rng(123)
    p = str(tmp_path / "fuzz.bin")
    outcomes = {"parsed": 0, "refused": 0}
    for trial in range(400):
        blob = bytearray(data)
        kind = trial % 4
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                blob[int(rng.integers(0, len(blob)))] = int(rng.integers(0, 256))
        elif kind == 1:
            blob = blob[:int(rng.integers(0, len(blob)))]
        elif kind == 2:
            pos = int(rng.integers(0, len(blob)))
            blob[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
        else:
            pos = int(rng.integers(30, len(blob) - 8))
            blob[pos:pos + 8] = bytes(rng.choice([0x00, 0x7f, 0x80, 0xff], 8).astype(np.uint8))
        open(p, "wb").write(bytes(blob))
        try:
            got = lio.read_segment_cache(p)
            assert got.segments.shape[1] == 4 and len(got.ci) == len(got.cj) == len(got.cw)
            outcomes["parsed"] += 1
        except RuntimeError as e:
            assert str(e)
            outcomes["refused"] += 1
    assert outcomes["refused"] > 200
    # the row count that wraps a 64-bit byte count
    hdr = data.find(struct.pack("<III", 4, 30, 8))
    assert hdr > 0
    blob = bytearray(data[:hdr + 44])
    blob[hdr + 4:hdr + 12] = struct.pack("<II", 0x80000000, 0x80000000)
    blob[hdr + 12:hdr + 28] = struct.pack("<QQ", 0x80000000 * 4, 0x80000000)
    open(p, "wb").write(bytes(blob))
    with pytest.raises(RuntimeError):
        lio.read_segment_cache(p)
