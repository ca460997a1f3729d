"""4-image mosaic on the device (reference utils/datasets_ssod.py:732-792 load_mosaic_with_M).

tests/golden/mosaic.npz comes from the LIVE reference (oracle/make_golden.py::case_mosaic): its own load_mosaic_with_M on a six-image
in-memory dataset, four mosaics from one seeded `random` stream -- the 2s x 2s canvas it pastes, the labels it hands on, the stream
position afterwards.  Pinned here: the host-side sampling consumes the stream exactly as the reference does, the placement
reproduces the reference's canvas pixel for pixel, the label arithmetic (including its clip to the CANVAS size) is equal.  The 2:1
resampling (cv2.resize at this ratio = OpenCV's INTER_AREA fast path, (a + b + c + d + 2) >> 2) is a restatement: cv2 is not
installed, so that step is unpinned; the kernel is compared with numpy's box average of the reference's canvas."""
import numpy as np
import torch

from tests.conftest import golden


def _dataset(g):
    imgs, labels = [], []
    k = 0
    while f"img{k}" in g.files:
        imgs.append(g[f"img{k}"]); labels.append(g[f"lab{k}"]); k += 1
    return imgs, labels


def test_mosaic_sampling_layout_and_labels_equal_the_reference():
    from efficientteacher_amd.utils.augment import MosaicGenerator, mosaic_labels, mosaic_layout
    g = golden("mosaic")
    s = int(g["s"][0])
    imgs, labels = _dataset(g)
    gen = MosaicGenerator(s, seed=1234)
    for j, index in enumerate(g["index"]):
        yc, xc, four = gen.sample(int(index), list(range(len(imgs))))
        shapes = [imgs[i].shape[:2] for i in four]
        layout = mosaic_layout(s, yc, xc, shapes)
        canvas = np.full((2 * s, 2 * s, 3), 114, np.uint8)                       # what the layout means, in numpy
        for (x1a, y1a, x2a, y2a, x1b, y1b, _, _), i in zip(layout, four):
            canvas[y1a:y2a, x1a:x2a] = imgs[i][y1b:y1b + (y2a - y1a), x1b:x1b + (x2a - x1a)]
        assert np.array_equal(canvas, g[f"canvas{j}"]), j                       # the reference's own canvas
        lab = mosaic_labels(s, layout, shapes, [labels[i] for i in four])
        assert lab.shape == g[f"labels4_{j}"].shape and np.allclose(lab, g[f"labels4_{j}"], rtol=1e-6, atol=1e-5), j
    assert gen.rng.random() == float(g["next_random"][0])                        # the `random` stream is where the reference left it


def test_mosaic_kernel_is_the_box_average_of_the_reference_canvas(hip):
    from efficientteacher_amd.utils.augment import MosaicGenerator
    g = golden("mosaic")
    s = int(g["s"][0])
    imgs, labels = _dataset(g)
    dev_imgs = [hip.t(np.ascontiguousarray(im.transpose(2, 0, 1))) for im in imgs]
    gen = MosaicGenerator(s, seed=1234)
    for j, index in enumerate(g["index"]):
        out, lab = gen(dev_imgs, labels, int(index))
        a = g[f"canvas{j}"].astype(np.int64)
        want = ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        assert np.array_equal(out.cpu().numpy().transpose(1, 2, 0), want), j
        assert np.array_equal(want, g[f"weak{j}"])                               # (what the golden run's recorder returned)
        assert np.allclose(lab, g[f"labels4_{j}"], rtol=1e-6, atol=1e-5)


def test_mosaic_batch_and_edges(hip):
    """several mosaics in one launch; a tile that does not reach the canvas edge leaves border pixels; odd tile sizes"""
    from efficientteacher_amd import ops
    from efficientteacher_amd.utils.augment import mosaic_layout
    rng = np.random.default_rng(2)
    s = 16
    tiles, layouts, want = [], [], []
    for b, (yc, xc) in enumerate(((8, 24), (23, 9), (16, 16))):
        shapes = [(int(rng.integers(3, s + 1)), int(rng.integers(3, s + 1))) for _ in range(4)]
        ims = [rng.integers(0, 256, (3, h, w), dtype=np.uint8) for h, w in shapes]
        lay = mosaic_layout(s, yc, xc, shapes)
        canvas = np.full((3, 2 * s, 2 * s), 114, np.int64)
        for (x1a, y1a, x2a, y2a, x1b, y1b, _, _), im in zip(lay, ims):
            canvas[:, y1a:y2a, x1a:x2a] = im[:, y1b:y1b + (y2a - y1a), x1b:x1b + (x2a - x1a)]
        want.append(((canvas[:, 0::2, 0::2] + canvas[:, 0::2, 1::2] + canvas[:, 1::2, 0::2] + canvas[:, 1::2, 1::2] + 2) >> 2).astype(np.uint8))
        tiles.append([hip.t(im) for im in ims]); layouts.append(lay)
    out = ops.mosaic4_u8(tiles, layouts, s).cpu().numpy()
    assert out.shape == (3, 3, s, s) and np.array_equal(out, np.stack(want))
