"""CPU tests of the visualisation helpers (generative_models_amd/viz.py; ns_gan.py:228-262)."""
import struct
import zlib

import numpy as np

from generative_models_amd import viz


def test_make_grid_layout_matches_torchvision_defaults():
    imgs = np.stack([np.full((4, 5), (k + 1) / 10.0, dtype=np.float32) for k in range(5)])
    g = viz.make_grid(imgs, nrow=3)
    assert g.shape == (2 * (4 + 2) + 2, 3 * (5 + 2) + 2)            # padding 2 around every tile
    assert g[0].sum() == 0 and g[:, 0].sum() == 0                    # pad value 0
    assert np.allclose(g[2:6, 2:7], 0.1) and np.allclose(g[2:6, 9:14], 0.2)
    assert np.allclose(g[8:12, 9:14], 0.5) and g[8:12, 16:21].sum() == 0   # 5 images: last cell empty


def test_png_roundtrip(tmp_path):
    img = np.linspace(0, 1, 12 * 7, dtype=np.float32).reshape(12, 7)
    p = tmp_path / "x.png"
    viz.write_png_gray(str(p), img)
    b = p.read_bytes()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    w, h, depth, ctype = struct.unpack(">IIBB", b[16:26])
    assert (w, h, depth, ctype) == (7, 12, 8, 0)
    i = b.index(b"IDAT")
    n = struct.unpack(">I", b[i - 4:i])[0]
    raw = zlib.decompress(b[i + 4:i + 4 + n])
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(12, 8)
    assert (rows[:, 0] == 0).all()
    want = np.clip(img * 255.0 + 0.5, 0, 255).astype(np.uint8)       # save_image's quantisation
    assert np.array_equal(rows[:, 1:], want)
