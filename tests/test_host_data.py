"""Host-side rows next to the hot path against vectors produced by the
reference's own modules (oracle/gen_golden_host.py -> tests/golden/host.npz)."""
import io
import os

import numpy as np
import pytest

from spr_pick_amd import coordinates, micrograph_io, sampler

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "host.npz"))


# ---- label rasterisers ----------------------------------------------------------------------
def test_as_mask_matches_reference():
    shape = tuple(GOLD["ras_shape"])
    xs, ys = GOLD["ras_x"], GOLD["ras_y"]
    for r, key in ((3, "ras_mask_r3"), (0, "ras_mask_r0")):
        got = coordinates.as_mask(shape, xs, ys, [r] * len(xs))
        assert got.dtype == np.uint8
        np.testing.assert_array_equal(got, GOLD[key])


def test_as_gaussian_matches_reference():
    shape = tuple(GOLD["ras_shape"])
    xs, ys = GOLD["ras_x"], GOLD["ras_y"]
    for bb, key in ((24, "ras_hm_bb24"), (36, "ras_hm_bb36")):
        got = coordinates.as_gaussian(shape, xs, ys, bb=bb)
        assert got.dtype == np.float32
        np.testing.assert_array_equal(got, GOLD[key])
    np.testing.assert_array_equal(coordinates.as_gaussian(shape, xs[:0], ys[:0], bb=24), GOLD["ras_hm_empty"])
    assert (GOLD["ras_hm_empty"] == -1).all()


def test_gaussian_radius_and_kernel():
    got = [coordinates.gaussian_radius((b, b)) for b in (12, 24, 32, 36)]
    np.testing.assert_array_equal(np.array(got), GOLD["gaussian_radius"])
    assert int(coordinates.gaussian_radius((24, 24))) == 6          # radius 6 for bb 24
    np.testing.assert_array_equal(coordinates.gaussian2d((13, 13), sigma=13 / 6), GOLD["gaussian2d_13"])


def test_coordinate_table_and_matching(tmp_path):
    p = tmp_path / "particles.txt"
    p.write_text("image_name\tx_coord\ty_coord\nmicA\t10\t12\nmicA\t50\t60\nmicB\t5\t7\n")
    table = coordinates.read_coordinates(str(p))
    by = coordinates.coordinates_by_image(table)
    np.testing.assert_array_equal(by[0]["micA"], np.array([[10, 12], [50, 60]], dtype=np.int32))
    images = {0: {"micA": np.zeros((120, 100), np.uint8), "micC": np.zeros((40, 30), np.uint8)}}
    m = coordinates.match_coordinates_to_images(table, images, radius=3, bb=24)
    assert list(m[0]) == ["micA", "micC"]
    _, mask, hm = m[0]["micA"]
    assert mask[12, 10] == 1 and mask[60, 50] == 1 and mask.sum() == 2 * 29      # closed disc r=3 has 29 pixels
    assert hm[12, 10] == 1.0 and hm[0, 99] == -1.0
    _, mask_c, hm_c = m[0]["micC"]
    assert mask_c.sum() == 0 and (hm_c == -1).all()
    with pytest.raises(NotImplementedError):
        coordinates.read_coordinates("x.csv")


# ---- sampler ----------------------------------------------------------------------------------
def _groups():
    return [[GOLD["smp_label_0_0"], GOLD["smp_label_0_1"]], [GOLD["smp_label_1_0"]]]


def test_enumerate_pu_matches_reference_loop():
    for g, group in enumerate(_groups()):
        P, U = sampler.enumerate_pu_coordinates(group)
        for got, key in ((P, "smp_P_%d" % g), (U, "smp_U_%d" % g)):
            pairs = np.stack([got >> np.uint64(32), got & np.uint64(0xFFFFFFFF)], 1).astype(np.int64)
            np.testing.assert_array_equal(pairs, GOLD[key])
        assert len(P) > 0 and len(U) > len(P)


def test_margin_rule_is_the_swapped_one():
    y = np.zeros((230, 250), np.uint8)
    _, U = sampler.enumerate_pu_coordinates([y])
    coord = (U & np.uint64(0xFFFFFFFF)).astype(np.int64)
    rows, cols = coord // 250, coord % 250
    assert rows.min() == 73 and rows.max() == 250 - 141      # rows are bounded by the COLUMN count
    assert cols.min() == 73 and cols.max() == 230 - 141      # and columns by the row count


@pytest.mark.parametrize("balance,tag", [(0.1, "b10"), (None, "bnone")])
def test_stratified_stream_is_the_reference_stream(balance, tag):
    s = sampler.StratifiedCoordinateSampler(_groups(), balance=balance, size=400, random=np.random.RandomState(7))
    np.testing.assert_array_equal(s.weights, GOLD["smp_weights_" + tag])
    draws = np.array([next(s) for _ in range(400)], dtype=np.int64)
    np.testing.assert_array_equal(draws, GOLD["smp_draws_" + tag])
    g, i, c = sampler.decode_index(int(draws[0]))
    assert g in (0, 1) and i in (0, 1) and 0 <= c < 250 * 230


def test_sampler_default_size_and_iteration():
    s = sampler.StratifiedCoordinateSampler(_groups(), balance=0.1, random=np.random.RandomState(7))
    assert len(s) == int(GOLD["smp_default_size"])
    assert len(list(iter(sampler.StratifiedCoordinateSampler(_groups(), balance=0.1, size=5,
                                                             random=np.random.RandomState(1))))) == 5
    assert sampler.sequential_indices(3, 7) == [0, 1, 2, 0, 1, 2, 0]


def test_sampler_rejects_micrographs_without_candidates():
    with pytest.raises(ValueError):
        sampler.StratifiedCoordinateSampler([[np.zeros((100, 100), np.uint8)]], balance=0.1)


# ---- MRC --------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [0, 1, 2, 6])
def test_mrc_parse_matches_reference(mode):
    arr, hdr, ext = micrograph_io.parse_mrc(GOLD["mrc%d_bytes" % mode].tobytes())
    want = GOLD["mrc%d_array" % mode]
    assert arr.dtype == want.dtype and arr.shape == want.shape
    np.testing.assert_array_equal(arr, want)
    assert hdr.mode == mode
    if mode == 2:
        nums = [hdr.nx, hdr.ny, hdr.nz, hdr.mode, hdr.next, hdr.amin, hdr.amax, hdr.amean, hdr.rms]
        np.testing.assert_array_equal(np.array(nums, dtype=np.float64), GOLD["mrc2_header_nums"])
    else:
        assert ext == bytes(range(16))


def test_mrc_stack_and_writer_roundtrip():
    arr, hdr, _ = micrograph_io.parse_mrc(GOLD["mrc_stack_bytes"].tobytes())
    np.testing.assert_array_equal(arr, GOLD["mrc_stack_array"])
    assert arr.shape == (2, 3, 4)
    buf = io.BytesIO()
    micrograph_io.write_mrc(buf, GOLD["mrc2_array"])
    assert buf.getvalue() == GOLD["mrc2_bytes"].tobytes()       # byte-identical to the reference writer
    with pytest.raises(ValueError):
        micrograph_io.parse_mrc(b"\x00" * 100)
    bad = bytearray(GOLD["mrc2_bytes"].tobytes())
    bad[12:16] = (4).to_bytes(4, "little")
    with pytest.raises(ValueError):
        micrograph_io.parse_mrc(bytes(bad))


def test_load_image_formats(tmp_path):
    from PIL import Image
    rng = np.random.RandomState(0)
    x = rng.randn(24, 40).astype(np.float32)
    with open(tmp_path / "m.mrc", "wb") as f:
        micrograph_io.write_mrc(f, x)
    u8 = micrograph_io.load_image(str(tmp_path / "m.mrc"))
    assert u8.dtype == np.uint8 and u8.shape == (24, 40)
    assert u8.min() == 0 and u8.max() in (254, 255)
    want = ((x - x.min()) / (x.max() - x.min()) * 255)
    assert np.abs(u8.astype(np.float64) - np.floor(want)).max() <= 1
    q = rng.randint(0, 256, size=(24, 40)).astype(np.uint8)
    Image.fromarray(q, "L").save(tmp_path / "m.png")
    f32 = micrograph_io.load_image(str(tmp_path / "m.png"))
    assert f32.dtype == np.float32
    np.testing.assert_array_equal(f32, q.astype(np.float32) * 6 / 255 - 3)
    Image.fromarray(q, "L").save(tmp_path / "m.tiff")
    np.testing.assert_array_equal(micrograph_io.load_image(str(tmp_path / "m.tiff")), q)
    np.testing.assert_array_equal(micrograph_io.to_unit_float(q), q.astype(np.float32) / np.float32(255))
    (tmp_path / "list.txt").write_text("image_name\tpath\nm\t%s\n" % (tmp_path / "m.mrc"))
    assert micrograph_io.read_image_table(str(tmp_path / "list.txt")) == [(0, "m", str(tmp_path / "m.mrc"))]
    names = sorted(n for _, n, _ in micrograph_io.read_image_table(str(tmp_path)))
    assert names == ["m", "m", "m"]


# ---- STAR / box tables and exporters ------------------------------------------------------------
def test_star_and_box_readers_match_reference(tmp_path):
    p = tmp_path / "p.star"
    p.write_bytes(GOLD["star_text"].tobytes())
    t = coordinates.read_coordinates(str(p))
    assert list(t.columns) == list(GOLD["star_columns"])
    np.testing.assert_array_equal(t[["x_coord", "y_coord"]].values.astype(np.int64), GOLD["star_xy"])
    np.testing.assert_array_equal(t["score"].values.astype(np.float64), GOLD["star_score"])
    assert list(t["image_name"]) == list(GOLD["star_names"])
    b = tmp_path / "micZ.box"
    b.write_text("10 20 30 40\n  5   6   7   9 extra\n")
    bt = coordinates.read_coordinates(str(b))
    np.testing.assert_array_equal(bt[["x_coord", "y_coord"]].values.astype(np.int64), GOLD["box_xy"])
    assert list(bt["image_name"]) == list(GOLD["box_names"])
    with pytest.raises(NotImplementedError):
        coordinates.read_coordinates("x.csv")
    with pytest.raises(ValueError):
        coordinates.read_coordinates("x.xyz")


def test_star_writer_and_score_export(tmp_path):
    import pandas as pd
    from spr_pick_amd import export
    table = pd.DataFrame({"image_name": ["m1", "m2"], "x_coord": [3, 4], "y_coord": [5, 6], "score": [0.25, 1.5]})
    out = io.StringIO()
    export.write_star(export.coordinates_to_star(table, image_ext=".mrc"), out)
    assert out.getvalue().encode() == GOLD["star_written"].tobytes()
    # convert_to_star.py: score > thr, strictly inside the window, x4, name = file name minus 18 chars + .mrc
    f = tmp_path / "micA_000064_scores.txt"          # an 18-character suffix, the script's constant
    f.write_text("image_name\tx_coord\ty_coord\tscore\nmicA\t100\t200\t0.5\nmicA\t15\t200\t0.9\nmicA\t300\t1009\t0.9\n"
                 "micA\t16\t16\t0.13\nmicA\t17\t18\t0.131\n")
    n = export.scores_to_star([str(f)], str(tmp_path / "o.star"))
    text = (tmp_path / "o.star").read_text()
    assert n == 2 and text.startswith(export.STAR_HEADER)
    assert text[len(export.STAR_HEADER):] == "400\t800\tmicA.mrc\t0.5\n68\t72\tmicA.mrc\t0.131\n"
    g = tmp_path / "micB_scores.txt"
    g.write_text("image_name\tx_coord\ty_coord\tscore\nmicB\t50\t60\t0.3\n")
    assert export.scores_to_star([str(g)], str(tmp_path / "o2.star"), strip=len("_scores.txt"), scale=1) == 1
    assert (tmp_path / "o2.star").read_text().endswith("50\t60\tmicB.mrc\t0.3\n")
    back = coordinates.read_coordinates(str(tmp_path / "o2.star"))          # and our reader takes it back
    assert list(back["image_name"]) == ["micB"] and int(back["x_coord"][0]) == 50 and float(back["score"][0]) == 0.3
