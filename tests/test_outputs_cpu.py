"""Host logic next to the hot path that needs no GPU: the PNG encoder / quantiser of the output writer against
the reference's statement (utils/data.py:71-93,143-147: min-max normalise, transpose back, uint8(x*255), PIL
save), the pick <-> planted-centre matching used by the full-pipeline recall / precision figures, and the
synthetic data set on disk."""
import io
import os

import numpy as np
import pytest
import torch


def _reference_png_pixels(img):
    """What save_tensor_image writes: NumPy float32 arithmetic of the reference, decoded back."""
    x = img.detach().to(torch.float32).cpu().numpy()
    lo, hi = float(x.min()), float(x.max())
    x = (x - lo) / (hi - lo) if hi > lo else np.zeros_like(x)
    return np.uint8(x[0].T * 255)


def test_quantise_and_png_match_the_reference_statement(tmp_path):
    from PIL import Image
    from spr_pick_amd import outputs
    g = torch.Generator().manual_seed(3)
    cases = [torch.rand(1, 37, 53, generator=g), torch.randn(1, 64, 64, generator=g) * 7 - 2,
             torch.full((1, 8, 9), 0.25), torch.rand(1, 1, 1, generator=g)]
    for k, img in enumerate(cases):
        want = _reference_png_pixels(img)
        got = outputs.quantise(img).numpy()
        np.testing.assert_array_equal(got, want)
        back = np.array(Image.open(io.BytesIO(outputs.png_bytes(got))))
        assert back.dtype == np.uint8 and back.shape == want.shape
        np.testing.assert_array_equal(back, want)
        path = str(tmp_path / ("t%d.png" % k))
        outputs.tensor_to_png(img, path)
        np.testing.assert_array_equal(np.array(Image.open(path)), want)


def test_output_writer_writes_everything_and_reraises(tmp_path):
    from PIL import Image
    from spr_pick_amd import outputs
    w = outputs.OutputWriter(threads=3)
    imgs = [torch.rand(1, 40, 30, generator=torch.Generator().manual_seed(i)) for i in range(20)]
    for i, im in enumerate(imgs):
        w.png(im, str(tmp_path / ("a%02d.png" % i)))
    w.call(lambda p: open(p, "w").write("x\n"), str(tmp_path / "s.txt"))
    w.drain()
    for i, im in enumerate(imgs):
        np.testing.assert_array_equal(np.array(Image.open(str(tmp_path / ("a%02d.png" % i)))), _reference_png_pixels(im))
    assert open(str(tmp_path / "s.txt")).read() == "x\n"
    w.png(imgs[0], str(tmp_path / "no_such_dir" / "a.png"))
    try:
        w.drain()
    except OSError:
        pass
    else:
        raise AssertionError("a failed write must surface in drain()")


def test_match_picks_and_metrics():
    from spr_pick_amd import picks
    truth = np.array([[10, 10], [50, 50], [90, 10]])
    xy = np.array([[11, 10], [52, 49], [200, 200], [12, 11], [90, 22]])
    s = np.array([0.9, 0.8, 0.7, 0.6, 0.3])
    order, hit = picks.match_picks(xy, s, truth, 12)
    assert order.tolist() == [0, 1, 2, 3, 4]
    # pick 3 is near centre 0, which pick 0 already took: one-to-one, so it is a false positive; pick 4 is exactly 12 away
    assert hit.tolist() == [True, True, False, False, True]
    m = picks.detection_metrics([(xy, s, truth)], 12, thresholds=(0.5, 0.75))
    assert m["n_truth"] == 3 and m["n_picks"] == 5
    assert m["at"][0.5] == {"precision": 0.5, "recall": 2 / 3, "picks": 4}
    assert m["at"][0.75] == {"precision": 1.0, "recall": 2 / 3, "picks": 2}
    assert abs(m["average_precision"] - (1 + 1 + 3 / 5) / 3) < 1e-12
    assert m["best_f1"]["picks"] == 2 and abs(m["best_f1"]["f1"] - 0.8) < 1e-12
    empty = picks.detection_metrics([(np.zeros((0, 2)), np.zeros(0), truth)], 12)
    assert empty["average_precision"] == 0.0 and empty["at"][0.5]["recall"] == 0.0
    # a closer free centre wins over a farther one
    o, h = picks.match_picks(np.array([[20, 10]]), np.array([1.0]), np.array([[10, 10], [24, 10]]), 12)
    assert h.tolist() == [True]


def test_scores_roundtrip_and_dataset_on_disk(tmp_path):
    from spr_pick_amd import coordinates, feed, micrograph_io, picks, synthetic
    p = str(tmp_path / "m_scores.txt")
    scores = np.array([0.9, 0.5, 0.4], dtype=np.float32)
    coords = np.array([[100, 40], [10, 50], [60, 70]], dtype=np.int32)        # (col, row); the second is in the border
    n = picks.write_scores(p, "m", scores, coords, (200, 200))
    names, xy, s = picks.read_scores(p)
    assert n == 2 and names == ["m", "m"] and xy.tolist() == [[40, 100], [70, 60]]
    assert [float(np.float32(v)) for v in s] == [float(scores[0]), float(scores[2])]

    ds = synthetic.write_dataset(str(tmp_path / "set"), 2, size=512, blobs=30)
    assert ds["n"] == 2 and ds["planted"] == 60 and 0 < ds["labelled"] < 60
    rows = micrograph_io.read_image_table(ds["images"])
    assert [r[1] for r in rows] == ["mic0000", "mic0001"] and all(os.path.exists(r[2]) for r in rows)
    q, centres, labelled = synthetic.micrograph(1, size=512, blobs=30)
    img = micrograph_io.load_image(rows[1][2])
    assert img.dtype == np.uint8 and img.shape == (512, 512) and np.abs(img.astype(int) - q.astype(int)).max() <= 1
    truth = coordinates.read_coordinates(ds["truth"])
    t1 = truth.loc[truth.image_name == "mic0001"]
    assert sorted(zip(t1.x_coord, t1.y_coord)) == sorted((int(cx), int(cy)) for cy, cx in centres)
    groups, names = feed.load_micrographs(ds["images"], ds["labels"], radius=3, bb=24)
    hm = groups[0][1][2]
    for cy, cx in labelled:
        assert hm[cy, cx] == 1.0
    assert (hm >= 0).sum() <= len(labelled) * 169


def test_png_dataset_is_the_standardised_image_on_the_loaders_png_path(tmp_path):
    """write_dataset(fmt="png"): 8-bit PNG levels of the standardised micrograph; the loader's PNG branch
    (utils/loader.py:72-82: unquantize to [-3, 3]) hands the network a zero-centred image with the same particles as
    the MRC variant of the same index."""
    from spr_pick_amd import coordinates, micrograph_io, synthetic
    ds = synthetic.write_dataset(str(tmp_path / "png"), 2, size=256, blobs=12, fmt="png")
    assert ds["format"] == "png" and ds["planted"] == 24
    rows = micrograph_io.read_image_table(ds["images"])
    assert all(r[2].endswith(".png") for r in rows)
    img = micrograph_io.load_image(rows[1][2])
    assert img.dtype == np.float32 and img.shape == (256, 256)
    assert abs(float(img.mean())) < 0.05 and 0.2 < float(img.std()) < 0.3 and -3.0 <= img.min() and img.max() <= 3.0
    q, centres, labelled = synthetic.micrograph_standardised(1, size=256, blobs=12)
    assert np.allclose(img, micrograph_io.unquantize(q))
    _, centres_mrc, labelled_mrc = synthetic.micrograph(1, size=256, blobs=12)
    assert np.array_equal(centres, centres_mrc) and np.array_equal(labelled, labelled_mrc)
    cy, cx = centres[0]
    assert img[cy, cx] < img.mean() - 0.2                      # a particle is a dark blob, 1.5 sigma deep
    truth = coordinates.read_coordinates(ds["truth"])
    assert len(truth) == 24
    with pytest.raises(ValueError):
        synthetic.write_dataset(str(tmp_path / "bad"), 1, size=256, fmt="tiff")


def test_metric_accumulates_the_per_sample_mean_in_one_reduction():
    """utils.Metric: mean over the non-sample axes, summed over the samples, divided by the sample count — the fused
    form (one sum + one add per step, an optional constant factor folded in) equals the three-operator definition."""
    import torch
    from spr_pick_amd import utils
    g = torch.Generator().manual_seed(3)
    vals = [torch.rand(4, 1, 8, 8, generator=g) for _ in range(3)]
    m = utils.Metric()
    for v in vals:
        m.add(v, scale=255.0)
    want = sum((v * 255).mean(dim=(1, 2, 3)).sum(dim=0) for v in vals) / 12
    assert m.n == 12 and torch.allclose(m.accumulated(), want, rtol=1e-5)
    s = utils.Metric()
    s += torch.tensor([2.0])                                   # one-element tensors (the scalar losses): no reduction at all
    s += torch.tensor([4.0])
    assert float(s.accumulated()) == 3.0 and s.n == 2
    keep = utils.Metric(collapse=False)                        # per-pixel metrics keep their shape
    keep += torch.ones(2, 3)
    keep += torch.zeros(2, 3)
    assert keep.accumulated().shape == (3,) and torch.allclose(keep.accumulated(), torch.full((3,), 0.5))
