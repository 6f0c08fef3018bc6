"""utils.get_data (utils.py:16-53) without the download: the IDX branch of the drop-in loader reads
MNIST files the user supplies, binarises them with the reference's seed-3435 Bernoulli protocol
(one draw per image, in dataset order) and splits the last 10 000 training images off as validation."""
import os
import struct

import numpy as np
import torch

from generative_models_amd.trainers import get_data


def _write_idx(path, arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(struct.pack(">BBBB", 0, 0, 8, arr.ndim))
        for d in arr.shape:
            f.write(struct.pack(">I", d))
        f.write(arr.tobytes())


def test_get_data_reads_idx_files_with_the_reference_protocol(tmp_path):
    rng = np.random.RandomState(0)
    n_tr, n_te = 10040, 30
    tr = rng.randint(0, 256, size=(n_tr, 28, 28)).astype(np.uint8)
    te = rng.randint(0, 256, size=(n_te, 28, 28)).astype(np.uint8)
    trl, tel = rng.randint(0, 10, n_tr).astype(np.uint8), rng.randint(0, 10, n_te).astype(np.uint8)
    raw = tmp_path / "MNIST" / "raw"
    os.makedirs(raw)
    _write_idx(raw / "train-images-idx3-ubyte", tr)
    _write_idx(raw / "t10k-images-idx3-ubyte", te)
    _write_idx(raw / "train-labels-idx1-ubyte", trl)
    _write_idx(raw / "t10k-labels-idx1-ubyte", tel)
    train_iter, val_iter, test_iter = get_data(BATCH_SIZE=16, root=str(tmp_path) + "/")
    # the reference's protocol, restated: seed, then bernoulli(ToTensor(image)) image by image
    torch.manual_seed(3435)
    f = lambda a: torch.from_numpy(a.astype(np.float32) / 255.0).view(-1, 1, 28, 28)
    ref_tr = torch.stack([torch.bernoulli(d) for d in f(tr)])
    ref_te = torch.stack([torch.bernoulli(d) for d in f(te)])
    timg, tlab = train_iter.dataset.tensors
    vimg, vlab = val_iter.dataset.tensors
    eimg, elab = test_iter.dataset.tensors
    assert timg.shape == (n_tr - 10000, 1, 28, 28) and vimg.shape == (10000, 1, 28, 28)
    assert torch.equal(timg, ref_tr[:-10000]) and torch.equal(vimg, ref_tr[-10000:])
    assert torch.equal(eimg, ref_te)
    assert torch.equal(tlab, torch.from_numpy(trl[:-10000].astype(np.int64)))
    assert torch.equal(vlab, torch.from_numpy(trl[-10000:].astype(np.int64)))
    assert torch.equal(elab, torch.from_numpy(tel.astype(np.int64)))
    assert set(torch.unique(timg).tolist()) <= {0.0, 1.0}          # binary: eligible for 1 bit / pixel
    for it in (train_iter, val_iter, test_iter):                   # shuffling loaders like utils.py:49-51
        assert isinstance(it.sampler, torch.utils.data.RandomSampler) and it.batch_size == 16


def test_get_data_without_files_is_synthetic_and_binary():
    a, b, c = get_data(BATCH_SIZE=8, root="/nonexistent/", n_train=64, n_val=16, n_test=16)
    assert a.dataset.tensors[0].shape == (64, 1, 28, 28)
    assert set(torch.unique(a.dataset.tensors[0]).tolist()) <= {0.0, 1.0}
