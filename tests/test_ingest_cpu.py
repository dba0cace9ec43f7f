"""Image-file side of the path on CPU: the numpy oracle of dataset.py:37-73 against independent restatements,
the file-list builders and index pairing of dataset.py:76-222, shards, and that the device path refuses to run without a GPU."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from PIL import Image

from discogan_modernized_amd import _lib
from discogan_modernized_amd import dataset as ds
from oracle import image_prep_ref as R


def _rand_image(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def test_oracle_sampling_convention_equals_torch_bilinear_half_pixel():
    """cv2.resize INTER_LINEAR = half-pixel centres + edge clamp, the convention of F.interpolate(bilinear, align_corners=False):
    an independent implementation of the same sampling rule must agree with the oracle's float form to fp32 rounding."""
    img = _rand_image(37, 53, 0)
    for S in (16, 20, 64, 74, 106):
        got = R.resize_linear_float(img, S)
        want = torch.nn.functional.interpolate(torch.tensor(img).permute(2, 0, 1)[None].double(), size=(S, S), mode="bilinear",
                                               align_corners=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(got - want).max() <= 255 * 4e-6, S
        # the 8-bit fixed-point form is the float form rounded to an integer, within one count
        assert np.abs(R.resize_linear_u8(img, S).astype(np.float64) - got).max() < 1.0, S
    # exact 2x reduction: the fixed-point form is the rounded 2x2 box mean (cv2 switches to its area kernel there: (a+b+c+d+2) >> 2)
    img = _rand_image(32, 32, 1)
    box = ((img.astype(np.int64).reshape(16, 2, 16, 2, 3).sum((1, 3)) + 2) >> 2).astype(np.uint8)
    assert np.array_equal(R.resize_linear_u8(img, 16), box)
    # identity size: nothing moves
    assert np.array_equal(R.resize_linear_u8(img, 32), img)


def test_oracle_erosion_equals_scipy_minimum_filter_and_the_reference_expression():
    import scipy.ndimage as ndi
    img = _rand_image(40, 256, 2)
    e = R.erode3x3(img)
    want = np.stack([ndi.minimum_filter(img[..., c], size=3, mode="constant", cval=255) for c in range(3)], -1)
    assert np.array_equal(e, want.astype(np.float64))
    # dataset.py:55-57 literally: 255 - dilate(255 - x) with a 3x3 ones kernel whose border pixels never win
    inv = 255.0 - img
    dil = np.stack([ndi.maximum_filter(inv[..., c], size=3, mode="constant", cval=-np.inf) for c in range(3)], -1)
    assert np.array_equal(255.0 - dil, e)


def test_oracle_crop_normalise_layout_are_exact():
    img = _rand_image(256, 512, 3)
    a, b = R.prepare_image(img, "A", 256), R.prepare_image(img, "B", 256)
    assert a.shape == b.shape == (3, 256, 256) and a.dtype == b.dtype == np.float32
    assert np.array_equal(b, (img[:, 256:].astype(np.float32) / 255.0).transpose(2, 0, 1))        # S = crop size: resize is the identity
    assert np.array_equal(a, (R.erode3x3(img[:, :256]).astype(np.float32) / 255.0).transpose(2, 0, 1))
    n = R.prepare_image(img[:, :256], None, 64)
    assert n.shape == (3, 64, 64) and 0.0 <= n.min() and n.max() <= 1.0
    assert np.abs(n * 255.0 - np.round(n * 255.0)).max() < 1e-4                                   # uint8 path: multiples of 1/255


def _write_tree(root):
    rng = np.random.default_rng(5)
    for split, n in (("train", 7), ("test", 3)):
        d = root / "edges2shoes" / split
        d.mkdir(parents=True)
        for i in range(n):
            Image.fromarray(rng.integers(0, 256, (256, 512, 3), dtype=np.uint8)).save(d / f"{i:03d}_AB.jpg", quality=95)
    for item, n in (("tops", 5), ("hanbok", 4)):
        for split in ("train", "test"):
            d = root / "custom" / item / split
            d.mkdir(parents=True)
            for i in range(n):
                Image.fromarray(rng.integers(0, 256, (40, 30, 3), dtype=np.uint8)).save(d / f"{item}{i}.png")
    d = root / "celebA" / "img_align_celeba"
    d.mkdir(parents=True)
    rows = ["6", "Male Smiling Young"]
    for i, (m, s, y) in enumerate([(1, 1, 1), (1, -1, 1), (-1, 1, -1), (-1, -1, 1), (1, 1, -1), (-1, 1, 1)]):
        rows.append(f"{i:06d}.jpg {m} {s} {y}")
    (root / "celebA" / "list_attr_celeba.txt").write_text("\n".join(rows) + "\n")


def test_file_lists_follow_the_reference_rules(tmp_path, monkeypatch):
    _write_tree(tmp_path)
    monkeypatch.setattr(ds, "dataset_path", tmp_path)
    monkeypatch.setattr(ds, "shoe_path", tmp_path / "edges2shoes")
    monkeypatch.setattr(ds, "handbag_path", tmp_path / "edges2handbags")
    monkeypatch.setattr(ds, "celebA_path", tmp_path / "celebA")
    monkeypatch.setattr(ds, "facescrub_path", tmp_path / "facescrub")
    a, b = ds.get_edge2photo_files("edges2shoes", test=False)
    assert a is b and len(a) == 7 and a == sorted(a) and all(p.endswith("_AB.jpg") for p in a)     # both domains: the same files
    assert len(ds.get_edge2photo_files("edges2shoes", test=True)[0]) == 3
    with pytest.raises(FileNotFoundError):
        ds.get_edge2photo_files("edges2handbags")
    with pytest.raises(ValueError):
        ds.get_edge2photo_files("edges2cats")
    ca, cb = ds.get_custom_data("tops", "hanbok", test=False)
    assert len(ca) == 5 and len(cb) == 4
    # celebA: style_A = '1' vs style_B = '1' (or style_A = '-1'), optional constraint, the LAST n_test are the test split
    ma, mb = ds.get_celebA_files("Male", None, None, None, test=False, n_test=1)
    assert [os.path.basename(p) for p in ma] == ["000000.jpg", "000001.jpg"] and [os.path.basename(p) for p in mb] == ["000002.jpg", "000003.jpg"]
    ta, tb = ds.get_celebA_files("Male", None, None, None, test=True, n_test=1)
    assert os.path.basename(ta[0]) == "000004.jpg" and os.path.basename(tb[0]) == "000005.jpg"
    sa, sb = ds.get_celebA_files("Male", "Smiling", "Young", "1", test=False, n_test=1)
    assert [os.path.basename(p) for p in sa] == ["000000.jpg"] and [os.path.basename(p) for p in sb] == ["000000.jpg"]
    args = SimpleNamespace(task_name="tops2hanbok", n_test=2, image_size=64, style_A=None, style_B=None, constraint=None, constraint_type=None)
    da, db, ta, tb = ds.get_data(args)
    assert len(da) == 5 and len(db) == 4 and len(ta) == 5
    args.task_name = "hanbok2tops"
    assert len(ds.get_data(args)[0]) == 4
    assert ds.task_domains("edges2shoes") == ("A", "B") and ds.task_domains("handbags2shoes") == ("B", "B") and ds.task_domains("celebA") == (None, None)
    # index pairing of DiscoGANDataset (dataset.py:210-222)
    dset = ds.DiscoGANDataset(list(ca), list(cb), None, None, image_size=64)
    assert len(dset) == 4 and dset.paths(3) == (ca[3], cb[3]) and dset.paths(6) == (ca[1], cb[2])
    ia, ib = dset[1]
    assert ia.shape == (40, 30, 3) and ia.dtype == np.uint8 and np.array_equal(ib, np.asarray(Image.open(cb[1]).convert("RGB")))
    # independent shuffles (dataset.py:24-35)
    x, y = ds.shuffle_data(np.arange(50), np.arange(50), np.random.default_rng(0))
    assert sorted(x) == sorted(y) == list(range(50)) and not np.array_equal(x, y)


def test_shards_round_trip(tmp_path):
    _write_tree(tmp_path)
    files = sorted(str(p) for p in (tmp_path / "edges2shoes" / "train").glob("*.jpg"))
    p1 = ds.write_shard(files[:4], str(tmp_path / "s1.npy"), workers=2)
    p2 = ds.write_shard(files[4:], str(tmp_path / "s2.npy"), workers=2)
    src = ds.ShardSource([p1, p2])
    assert len(src) == 7 and src.shape == (256, 512, 3)
    row = np.empty((256, 512, 3), np.uint8)
    for i in (0, 3, 4, 6):
        src.fetch(i, row)
        assert np.array_equal(row, ds.decode_rgb(files[i]))
    fsrc = ds.FileSource(files)
    fsrc.fetch(5, row)
    assert np.array_equal(row, ds.decode_rgb(files[5]))
    with pytest.raises(ValueError):
        ds.write_shard([files[0], str(next((tmp_path / "custom" / "tops" / "train").glob("*.png")))], str(tmp_path / "bad.npy"))


def test_device_path_has_no_cpu_fallback():
    with pytest.raises(_lib.DiscoganHipError):
        ds.prepare_batch(torch.zeros((2, 8, 8, 3), dtype=torch.uint8), None, 4)
    with pytest.raises(_lib.DiscoganHipError):
        ds.DeviceLoader(None, None, (None, None), 4, [], device="cpu")
    with pytest.raises(ValueError):
        ds._crop_of("A", 200)
    assert ds._crop_of("B", 512) == (256, 256, 0, 1) and ds._crop_of("A", 512) == (0, 256, 1, 0) and ds._crop_of(None, 178) == (0, 178, 0, 1)
