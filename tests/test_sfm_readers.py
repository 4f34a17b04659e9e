"""SfM front ends: the C++ readers behind the C ABI against the oracle's pure-Python readers on files written
from known cameras.  No GPU needed for the readers themselves."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import l3d_oracle_sfm as osfm  # noqa: E402
from helpers import synth_worldpoints, write_bundler, write_nvm, assert_lines_equal  # noqa: E402
from line3d_amd.synth import make_scene  # noqa: E402


@pytest.fixture(scope="module")
def sfm_scene():
    sc = make_scene(10, 300, 6, seed=77)
    return sc, synth_worldpoints(sc, 400, seed=5)


def _same(cams_a, cams_b):
    assert len(cams_a) == len(cams_b)
    for a, b in zip(cams_a, cams_b):
        assert a["name"] == b["name"] and a["focal"] == b["focal"]
        assert np.array_equal(a["dist"], b["dist"]) and np.array_equal(a["R"], b["R"]) and np.array_equal(a["t"], b["t"])
        assert np.array_equal(a["worldpoints"], b["worldpoints"])


def test_nvm_reader_matches_oracle_and_truth(sfm_scene, tmp_path):
    from line3d_amd import sfm
    sc, pts = sfm_scene
    path = str(tmp_path / "scene.nvm")
    write_nvm(path, sc, pts)
    got = sfm.read_nvm(path)
    exp, npts = osfm.read_nvm(path)
    assert got.n_points == npts == len(pts)
    _same(got.cameras, exp)
    for cam, v in zip(got.cameras, sc.views):
        assert np.allclose(cam["R"], v["R"], atol=1e-12) and np.allclose(cam["t"], v["t"], atol=1e-11)
        assert cam["focal"] == float(np.float32(v["K"][0, 0]))
        assert np.array_equal(sfm.intrinsics(cam["focal"], v["width"], v["height"]), osfm.intrinsics(cam["focal"], v["width"], v["height"]))
        assert np.array_equal(sfm.intrinsics(cam["focal"], v["width"], v["height"]), v["K"])
    assert sum(len(c["worldpoints"]) for c in got.cameras) == sum(len(o) for _, o in pts) > 1000


def test_bundler_reader_matches_oracle_and_truth(sfm_scene, tmp_path):
    from line3d_amd import sfm
    sc, pts = sfm_scene
    path = str(tmp_path / "bundle.rd.out")
    write_bundler(path, sc, pts)
    got = sfm.read_bundler(path)
    exp, npts = osfm.read_bundler(path)
    assert got.n_points == npts == len(pts)
    _same(got.cameras, exp)
    for i, (cam, v) in enumerate(zip(got.cameras, sc.views)):
        assert np.array_equal(cam["R"], v["R"]) and np.array_equal(cam["t"], v["t"]) and cam["name"] == "%08d" % i


def test_reader_errors(tmp_path):
    from line3d_amd import sfm
    with pytest.raises(RuntimeError, match="does not exist"):
        sfm.read_nvm(str(tmp_path / "missing.nvm"))
    p = str(tmp_path / "empty.nvm")
    open(p, "w").write("NVM_V3\n\n0\n")
    with pytest.raises(RuntimeError, match="No aligned cameras"):
        sfm.read_nvm(p)
    assert sfm.result_basename(neighbors=10) == "line3D_result__W_-1__N_10__tL_1__tU_5__sigmaP_3.5__sigmaA_10__COLLIN__NO_DIFFUSION"


@pytest.mark.gpu
def test_driver_flow_from_nvm_matches_oracle(sfm_scene, tmp_path):
    """main_vsfm's flow with segments in place of images: NVM -> addImage with world point lists (similarity from shared
    world points, line3D.cc:1874-1935) -> compute3Dmodel -> TXT/STL, against the oracle fed by its own reader."""
    import l3d_oracle_pipeline as op
    from line3d_amd import sfm
    from line3d_amd.io import load_txt
    sc, pts = sfm_scene
    path = str(tmp_path / "scene.nvm")
    write_nvm(path, sc, pts)
    scene = sfm.read_nvm(path)
    segs = [v["segments"] for v in sc.views]
    sizes = [(v["width"], v["height"]) for v in sc.views]
    l3d = sfm.reconstruct(scene, segs, sizes, out_dir=str(tmp_path / "out"), neighbors=6)
    cams, _ = osfm.read_nvm(path)
    o = op.OracleLine3D(matching_neighbors=6)
    for i, c in enumerate(cams):
        assert o.add_image(i, sizes[i][0], sizes[i][1], segs[i], osfm.intrinsics(c["focal"], *sizes[i]), c["R"], c["t"], list(c["worldpoints"]))
    o.compute3Dmodel(False)
    assert len(o.result) > 0
    assert_lines_equal(l3d.getResult(), o.result, 1e-4)
    txt = os.path.join(str(tmp_path / "out"), sfm.result_basename(neighbors=6) + ".txt")
    assert len(load_txt(txt)) == len(o.result) and os.path.exists(txt[:-4] + ".stl")
    l3d.close()


@pytest.mark.gpu
def test_driver_flow_replays_segment_caches(sfm_scene, tmp_path):
    """main_vsfm's flow when the data directory already holds the segment caches of an earlier run (line3D.cc:143-168):
    segments AND collinearities come from the files -- here thinned out, so that they differ from what would be computed --
    against the oracle replaying the same files through its own reader."""
    import l3d_oracle_pipeline as op
    from line3d_amd import sfm
    from line3d_amd.io import segment_cache_filename
    sc, pts = sfm_scene
    path = str(tmp_path / "scene.nvm")
    write_nvm(path, sc, pts)
    scene = sfm.read_nvm(path)
    sizes = [(v["width"], v["height"]) for v in sc.views]
    data_dir = str(tmp_path / "L3D_data")
    os.makedirs(data_dir)
    olib = op.load_lib()
    n_entries = n_kept = 0
    for i, v in enumerate(sc.views):
        rel = op.collinearity(olib, v["segments"], 2.0)
        coll = {}
        ii, jj = np.nonzero(np.triu(rel > 0.0, 1))
        for a, b in zip(ii.tolist(), jj.tolist()):
            n_entries += 1
            if (a * 7 + b * 3 + i) % 4 == 0:                    # the earlier run "saw" fewer collinear pairs
                continue
            n_kept += 1
            coll.setdefault(a, {})[b] = rel[b, a]
            coll.setdefault(b, {})[a] = rel[b, a]
        osfm.write_segment_cache(data_dir + segment_cache_filename(i, sizes[i][0], sizes[i][1], True), v["segments"], coll)
    assert 0 < n_kept < n_entries
    l3d = sfm.reconstruct(scene, data_dir, sizes, neighbors=6)
    cams, _ = osfm.read_nvm(path)
    o = op.OracleLine3D(matching_neighbors=6)
    for i, c in enumerate(cams):
        assert o.add_image_cached(i, sizes[i][0], sizes[i][1], data_dir + osfm.filename_segment_cache(i, sizes[i][0], sizes[i][1], True),
                                  osfm.intrinsics(c["focal"], *sizes[i]), c["R"], c["t"], list(c["worldpoints"]))
    o.compute3Dmodel(False)
    assert len(o.result) > 0
    assert_lines_equal(l3d.getResult(), o.result, 1e-4)
    l3d.close()
    with pytest.raises(RuntimeError, match="no segment cache"):
        sfm.reconstruct(scene, str(tmp_path / "nothing_here"), sizes, neighbors=6)


@pytest.mark.gpu
def test_cpp_driver_over_segment_caches_matches_oracle(sfm_scene, tmp_path):
    """examples/main_vsfm_amd.cpp -- the reference driver's flow in C++ over the facade, the NVM reader and the segment caches
    (image sizes read off the cache file names) -- built with plain g++, run as a program, its TXT result against the oracle."""
    import subprocess
    import l3d_oracle_pipeline as op
    from line3d_amd import sfm
    from line3d_amd.io import load_txt, segment_cache_filename
    sc, pts = sfm_scene
    nvm = str(tmp_path / "scene.nvm")
    write_nvm(nvm, sc, pts)
    sizes = [(v["width"], v["height"]) for v in sc.views]
    data_dir = str(tmp_path / "L3D_data")
    os.makedirs(data_dir)
    olib = op.load_lib()
    for i, v in enumerate(sc.views):
        rel = op.collinearity(olib, v["segments"], 2.0)
        coll = {}
        ii, jj = np.nonzero(np.triu(rel > 0.0, 1))
        for a, b in zip(ii.tolist(), jj.tolist()):
            coll.setdefault(a, {})[b] = rel[b, a]
            coll.setdefault(b, {})[a] = rel[b, a]
        osfm.write_segment_cache(data_dir + segment_cache_filename(i, sizes[i][0], sizes[i][1], True), v["segments"], coll)
    exe = str(tmp_path / "main_vsfm_amd")
    lib = os.path.join(ROOT, "line3d_amd")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "main_vsfm_amd.cpp"),
                           "-L" + lib, "-lline3d_amd", "-Wl,-rpath," + lib, "-o", exe])
    out_dir = str(tmp_path / "out")
    os.makedirs(out_dir)
    r = subprocess.run([exe, nvm, data_dir, "6", "0", out_dir], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    txt = os.path.join(out_dir, sfm.result_basename(neighbors=6) + ".txt")
    got = load_txt(txt)
    assert os.path.exists(txt[:-4] + ".stl")
    cams, _ = osfm.read_nvm(nvm)
    o = op.OracleLine3D(matching_neighbors=6)
    for i, c in enumerate(cams):
        assert o.add_image(i, sizes[i][0], sizes[i][1], sc.views[i]["segments"], osfm.intrinsics(c["focal"], *sizes[i]), c["R"], c["t"], list(c["worldpoints"]))
    o.compute3Dmodel(False)
    assert len(got) == len(o.result) > 0
    for (g2, g3), (o2, o3) in zip(got, o.result):
        assert [(c, s) for c, s, _ in g2] == [tuple(k) for k in o2]
        assert len(g3) == len(o3)
        for (gp, gq), (p, q) in zip(g3, o3):
            assert np.allclose(gp, p, rtol=0, atol=1e-4) and np.allclose(gq, q, rtol=0, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("max_width", [-1, 960])
def test_reference_signature_driver_replays_cached_scene(sfm_scene, tmp_path, max_width):
    """tests/cpp/driver_reference_signatures.cpp: main_vsfm's flow with the reference's OWN addImage signature (line3D.h:69-73: id, image, K, R, t,
    world points, maxImgWidth, loadAndStoreSegments) -- the image a cv::Mat-shaped size stub, cameras Eigen-shaped, segments from the segment caches
    under the name the maxImgWidth rule gives (line3D.cc:130-150).  Its TXT result against the oracle; its three guarded calls add nothing."""
    import subprocess
    import l3d_oracle_pipeline as op
    from line3d_amd import sfm
    from line3d_amd.io import load_txt, segment_cache_filename
    from test_facade_header import _build_driver
    sc, pts = sfm_scene
    nvm = str(tmp_path / "scene.nvm")
    write_nvm(nvm, sc, pts)
    img_dir, out_dir = str(tmp_path / "images"), str(tmp_path / "out")
    data_dir = out_dir + "/L3D_data"
    os.makedirs(img_dir)
    os.makedirs(data_dir)
    olib = op.load_lib()
    for i, v in enumerate(sc.views):
        w, h = v["width"], v["height"]
        open(os.path.join(img_dir, "img_%04d.jpg" % i), "w").write("%d %d\n" % (w, h))       # (what the cv::imread double reads)
        nw, nh = w, h
        if max_width > 0 and max(w, h) > max_width:                                          # line3D.cc:133-138
            scale = np.float32(max_width) / np.float32(max(w, h))
            nw, nh = int(np.round(np.float32(w) * scale)), int(np.round(np.float32(h) * scale))
        rel = op.collinearity(olib, v["segments"], 2.0)
        coll = {}
        ii, jj = np.nonzero(np.triu(rel > 0.0, 1))
        for a, b in zip(ii.tolist(), jj.tolist()):
            coll.setdefault(a, {})[b] = rel[b, a]
            coll.setdefault(b, {})[a] = rel[b, a]
        if i != len(sc.views) - 1 or max_width > 0:
            osfm.write_segment_cache(data_dir + segment_cache_filename(i, nw, nh, True), v["segments"], coll)
    # at the native size the LAST camera has no cache: the reference would detect its segments; here the call prints and returns, the view is missing
    n_added = len(sc.views) - (0 if max_width > 0 else 1)
    exe = _build_driver(str(tmp_path))
    r = subprocess.run([exe, nvm, img_dir, out_dir, "6", "0", str(max_width)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "#images:         %d" % n_added in r.stdout
    assert "imageID already in use!" in r.stderr and "image is empty!" in r.stderr and "no segment cache" in r.stderr
    assert "3D lines found!" in r.stdout and "#filtered_matches (2):" in r.stdout            # verbose = true: the reference's counters
    txt = os.path.join(out_dir, sfm.result_basename(neighbors=6).replace("W_-1", "W_%d" % max_width) + ".txt")
    got = load_txt(txt)
    cams, _ = osfm.read_nvm(nvm)
    o = op.OracleLine3D(matching_neighbors=6)
    for i, c in enumerate(cams[:n_added]):
        w, h = sc.views[i]["width"], sc.views[i]["height"]
        assert o.add_image(i, w, h, sc.views[i]["segments"], osfm.intrinsics(c["focal"], w, h), c["R"], c["t"], list(c["worldpoints"]))
    o.compute3Dmodel(False)
    assert len(got) == len(o.result) > 0
    for (g2, g3), (o2, o3) in zip(got, o.result):
        assert [(c, s) for c, s, _ in g2] == [tuple(k) for k in o2]
        assert len(g3) == len(o3)
        for (gp, gq), (p, q) in zip(g3, o3):
            assert np.allclose(gp, p, rtol=0, atol=1e-4) and np.allclose(gq, q, rtol=0, atol=1e-4)


def test_sfm_readers_survive_corrupted_files(sfm_scene, tmp_path):
    """Truncated and garbled NVM / bundler files: the readers return cameras or raise with a message; they never crash and never
    trust a count the file cannot back."""
    from line3d_amd import sfm
    sc, pts = sfm_scene
    rng = np.random.default_rng(7)
    for writer, reader, name in ((write_nvm, sfm.read_nvm, "scene.nvm"), (write_bundler, sfm.read_bundler, "bundle.rd.out")):
        good = str(tmp_path / name)
        writer(good, sc, pts)
        text = open(good).read()
        p = str(tmp_path / ("fuzz_" + name))
        outcomes = {"read": 0, "refused": 0}
        for trial in range(120):
            kind = trial % 4
            if kind == 0:
                t = text[:int(rng.integers(0, len(text)))]
            elif kind == 1:
                toks = text.split(" ")
                for _ in range(int(rng.integers(1, 8))):
                    toks[int(rng.integers(0, len(toks)))] = str(rng.choice(["-1", "999999999", "nan", "x", "1e400", "", "4294967296", "-2147483649"]))
                t = " ".join(toks)
            elif kind == 2:
                lines = text.split("\\n")
                del lines[int(rng.integers(0, len(lines)))]
                t = "\\n".join(lines)
            else:
                pos = int(rng.integers(0, len(text)))
                t = text[:pos] + "".join(chr(int(c)) for c in rng.integers(32, 127, int(rng.integers(1, 30)))) + text[pos:]
            open(p, "w").write(t)
            try:
                got = reader(p)
                assert len(got.cameras) >= 0
                outcomes["read"] += 1
            except RuntimeError as e:
                assert str(e)
                outcomes["refused"] += 1
        assert outcomes["read"] + outcomes["refused"] == 120


@pytest.mark.gpu
def test_add_image_writes_reads_and_removes_segment_caches(small_scene, small_oracle, tmp_path):
    """Line3D::addImage's cache behaviour (line3D.cc:128-199, loadAndStoreSegments): a first run writes one cache per view
    (segments + the collinearity relation, the bytes the oracle's writer produces), a second run reads them INSTEAD of the segments
    it is handed (here: garbage) and reconstructs the same lines, a run with loadAndStoreSegments = false removes them, and
    maxImgWidth decides the size in the file name (line3D.cc:133-150)."""
    import l3d_oracle_pipeline as op
    from line3d_amd.pipeline import Line3D
    from line3d_amd.io import segment_cache_filename
    d = str(tmp_path / "L3D_data")
    os.makedirs(d)

    def run(segments_of, load_and_store, max_w=1920):
        l = Line3D(d, matchingNeighbors=6)
        for v in small_scene.views:
            assert l.addImage_ex(v["id"], v["width"], v["height"], segments_of(v), v["K"], v["R"], v["t"], v["sims"], max_w, load_and_store, fixed_sim=True)
        l.compute3Dmodel(False)
        return l

    l1 = run(lambda v: v["segments"], True)
    assert_lines_equal(l1.getResult(), small_oracle.result, 1e-4)
    files = sorted(os.listdir(d))
    assert files == sorted(segment_cache_filename(v["id"], v["width"], v["height"], True)[1:] for v in small_scene.views)
    olib = op.load_lib()
    for v in small_scene.views:                                   # the oracle's writer on the oracle's relation: the same bytes
        rel = op.collinearity(olib, v["segments"], 2.0)
        coll = {}
        ii, jj = np.nonzero(np.triu(rel > 0.0, 1))
        for a, b in zip(ii.tolist(), jj.tolist()):
            coll.setdefault(a, {})[b] = rel[b, a]
            coll.setdefault(b, {})[a] = rel[b, a]
        ref = str(tmp_path / "ref.bin")
        osfm.write_segment_cache(ref, v["segments"], coll, library_version=17)
        got = open(d + segment_cache_filename(v["id"], v["width"], v["height"], True), "rb").read()
        assert got == open(ref, "rb").read()
    l1.close()
    junk = np.array([[1.0, 2.0, 30.0, 40.0]] * 7, np.float32)
    l2 = run(lambda v: junk, True)                               # the caches stand in for the segments
    assert_lines_equal(l2.getResult(), small_oracle.result, 1e-4)
    l2.close()
    l3 = run(lambda v: v["segments"], False)                     # loadAndStoreSegments = false: the files are removed, nothing is written
    assert os.listdir(d) == []
    assert_lines_equal(l3.getResult(), small_oracle.result, 1e-4)
    l3.close()
    l4 = run(lambda v: v["segments"], True, max_w=960)           # the detector would have worked at half size: that size names the file
    v0 = small_scene.views[0]
    assert segment_cache_filename(v0["id"], 960, 540, True)[1:] in os.listdir(d)
    assert_lines_equal(l4.getResult(), small_oracle.result, 1e-4)
    l4.close()
