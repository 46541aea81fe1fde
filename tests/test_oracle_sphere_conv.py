"""Pins oracle.sphere_conv against fixtures generated from the REFERENCE class
(tests/golden/make_golden.py, ladcast/models/sphere_conv.py) and its docstring KAT."""
import os

import numpy as np
import torch

from oracle.sphere_conv import SphereConv2d, sphere_pad


def _load(golden_dir):
    return np.load(os.path.join(golden_dir, "sphere_conv_ref.npz"))


def test_docstring_kat(golden_dir):
    g = _load(golden_dir)
    # models/sphere_conv.py:142-172, typed in by hand here as well as taken from the fixture
    want = torch.tensor(
        [[44.0, 48, 52, 40, 44, 48, 52, 40], [48, 44, 48, 44, 48, 44, 48, 44], [52, 40, 44, 48, 52, 40, 44, 48]]
    )
    want_pad = torch.tensor(
        [
            [10, 11, 12, 13, 14, 15, 8, 9, 10, 11, 12, 13],
            [2, 3, 4, 5, 6, 7, 0, 1, 2, 3, 4, 5],
            [6, 7, 0, 1, 2, 3, 4, 5, 6, 7, 0, 1],
            [14, 15, 8, 9, 10, 11, 12, 13, 14, 15, 8, 9],
            [22, 23, 16, 17, 18, 19, 20, 21, 22, 23, 16, 17],
            [18, 19, 20, 21, 22, 23, 16, 17, 18, 19, 20, 21],
            [10, 11, 12, 13, 14, 15, 8, 9, 10, 11, 12, 13],
        ]
    )
    assert torch.equal(torch.from_numpy(g["kat_y"])[0, 0], want)
    assert torch.equal(torch.from_numpy(g["kat_pad"])[0, 0], want_pad)
    tmp = torch.arange(0, 24).view(1, 1, 3, 8)
    assert torch.equal(sphere_pad(tmp, (2, 2))[0, 0], want_pad)
    c = SphereConv2d(1, 1, 5, 1, 2)
    c.weight.data = torch.tensor(
        [[[[0, 1, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 1, 0], [0, 0, 0, 1, 0]]]], dtype=torch.float32
    )
    c.bias.data = torch.tensor([0.0])
    with torch.no_grad():
        assert torch.equal(c(tmp.float())[0, 0], want)


def test_matches_reference_fixtures(golden_dir):
    g = _load(golden_dir)
    i = 0
    while f"c{i}_meta" in g:
        ci, co, k, grp, b, H, W = [int(v) for v in g[f"c{i}_meta"]]
        m = SphereConv2d(ci, co, k, 1, k // 2, groups=grp, bias=bool(b))
        m.weight.data = torch.from_numpy(g[f"c{i}_w"])
        if b:
            m.bias.data = torch.from_numpy(g[f"c{i}_b"])
        x = torch.from_numpy(g[f"c{i}_x"])
        with torch.no_grad():
            y = m(x)
        assert torch.equal(sphere_pad(x, (k // 2, k // 2)), torch.from_numpy(g[f"c{i}_pad"]))
        # same torch conv2d on the same slices -> bit-exact
        assert torch.equal(y, torch.from_numpy(g[f"c{i}_y"])), f"case {i}"
        i += 1
    assert i == 5
