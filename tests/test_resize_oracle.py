"""CPU tests of the preprocessing oracle (oracle/resize_ref.py) and the host-side width rules.

cv2 is not installed, so the INTER_AREA restatement is pinned only by known answers that follow from the
published algorithm (and, for BGR2GRAY, by OpenCV's documented outputs for pure red / green / blue);
"parity unpinned" against real cv2 output, as the oracle's header says."""
import numpy as np
import pytest

from oracle import resize_ref as R


def test_gray_known_answers():
    # cv2.cvtColor(BGR2GRAY) of pure blue / green / red is 29 / 150 / 76
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0]]], np.uint8)
    assert R.bgr2gray(px, "bgr").tolist() == [[29, 150, 76, 255, 0]]
    assert R.bgr2gray(px[..., ::-1], "rgb").tolist() == [[29, 150, 76, 255, 0]]
    g = np.arange(256, dtype=np.uint8).reshape(16, 16)
    assert np.array_equal(R.bgr2gray(np.stack([g, g, g], axis=2)), g)      # coefficients sum to 2^14


def test_width_rules():
    assert R.target_width(48, 1318) == 3514 and R.target_width(53, 376) == 908      # SURVEY C1 widths
    assert R.target_width(57, 1058) == 2375 and R.target_width(77, 1151) == 1913 and R.target_width(54, 206) == 488
    assert R.target_width(300, 1000, rule="dataset") == int(1000 * (128 / 300))
    assert R.align_collate_widths([100, 2000, 50]) == 1600 and R.align_collate_widths([100, 900]) == 900
    assert R.truncate_label("abcdefghij", 2000, 1600) == "abcdefgh" and R.truncate_label("ab", 100, 1600) == "ab"
    assert R.truncate_label("a", 5000, 1600) == "a"


def test_area_identity_and_integer_decimation():
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (128, 77), dtype=np.uint8)
    assert np.array_equal(R.resize_area(a, 77, 128), a)
    b = rng.integers(0, 256, (256, 154), dtype=np.uint8).astype(np.int64)
    want = (b[0::2, 0::2] + b[0::2, 1::2] + b[1::2, 0::2] + b[1::2, 1::2] + 2) >> 2
    assert np.array_equal(R.resize_area(b.astype(np.uint8), 77, 128), want)
    c = rng.integers(0, 256, (384, 30), dtype=np.uint8)
    want = np.rint((c.reshape(128, 3, 10, 3).astype(np.float32).sum(axis=(1, 3)) * (np.float32(1) / np.float32(9)))
                   .astype(np.float64))
    assert np.array_equal(R.resize_area(c, 10, 128), want.astype(np.uint8))


def test_area_fractional_known_answer():
    # scale 1.5: cells [0,1.5) and [1.5,3): weights (2/3, 1/3) and (1/3, 2/3)
    assert R.area_tab(3, 2, 1.5) == [[(0, np.float32(1 / 1.5)), (1, np.float32(0.5 / 1.5))],
                                     [(1, np.float32(0.5 / 1.5)), (2, np.float32(1 / 1.5))]]
    src = np.array([[10, 20, 30]] * 3, np.uint8)
    assert R.resize_area(src, 2, 2).tolist() == [[13, 27], [13, 27]]
    # weights of every destination cell sum to 1 (to float32 rounding) for awkward ratios
    for ssize, dsize in ((1318, 1000), (77, 64), (129, 128), (1000, 3)):
        for ent in R.area_tab(ssize, dsize, ssize / dsize):
            assert abs(sum(float(a) for _, a in ent) - 1.0) < 1e-5


def test_enlarging_known_answers():
    # INTER_AREA enlarging by an integer factor replicates pixels
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (64, 40), dtype=np.uint8)
    assert np.array_equal(R.resize_area(a, 80, 128), np.repeat(np.repeat(a, 2, axis=0), 2, axis=1))
    assert np.array_equal(R.resize_area(a[:32], 160, 128), np.repeat(np.repeat(a[:32], 4, axis=0), 4, axis=1))
    # 2 -> 3 columns: dx=1 has sx=0, fx = 2 - 1*1.5 = 0.5 -> (1024, 1024)
    ofs, c0, c1 = R.linear_area_coeffs(2, 3, 2 / 3, 1.5, True)
    assert ofs.tolist() == [0, 0, 1] and c0.tolist() == [2048, 1024, 2048] and c1.tolist() == [0, 1024, 0]
    row = np.array([[0, 200]] * 2, np.uint8)
    assert R.resize_area(row, 3, 2).tolist() == [[0, 100, 200], [0, 100, 200]]


@pytest.mark.parametrize("h,w", [(48, 131), (77, 115), (127, 90), (129, 300), (200, 777), (300, 41), (1, 9), (500, 3)])
def test_constant_images_stay_constant_and_range(h, w):
    tw = max(1, R.target_width(h, w))
    for v in (0, 77, 255):
        assert np.unique(R.resize_area(np.full((h, w), v, np.uint8), tw, 128)).tolist() == [v]
    rng = np.random.default_rng(h * 1000 + w)
    a = rng.integers(0, 256, (h, w), dtype=np.uint8)
    o = R.resize_area(a, tw, 128)
    assert o.shape == (128, tw) and o.min() >= a.min() and o.max() <= a.max()
    assert abs(float(o.mean()) - float(a.mean())) < 12.0


def test_host_width_rules_match_oracle(pkg):
    pp = pkg.preprocess
    for h, w in ((48, 1318), (53, 376), (300, 1000), (128, 5), (129, 7), (1, 1)):
        for rule in ("test", "dataset"):
            assert pp.target_width(h, w, 128, rule) == R.target_width(h, w, 128, rule)
    assert pp.truncate_label("abcdefghij", 2000, 1600) == R.truncate_label("abcdefghij", 2000, 1600)
    with pytest.raises(ValueError):
        pp.target_width(1, 1, 128, "other")


def test_integer_decimation_agrees_with_an_independent_box_filter():
    """Not a cv2 pin (PIL is a different library with its own rounding), but an independent implementation of the
    same box average: for integer factors PIL's BOX resampling and the oracle differ by at most one grey level."""
    from PIL import Image
    rng = np.random.default_rng(3)
    for h, w in ((256, 600), (384, 900), (512, 64)):
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        tw = R.target_width(h, w)
        o = R.resize_area(a, tw, 128).astype(int)
        p = np.asarray(Image.fromarray(a).resize((tw, 128), Image.BOX)).astype(int)
        assert o.shape == p.shape and np.abs(o - p).max() <= 1
