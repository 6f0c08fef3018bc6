"""Visualisation path of the reference trainers (ns_gan.py:228-281: generate_images, viz_loss;
SURVEY.md 8f item 4), made device-safe and kept OFF the step stream: it runs at epoch end, after the
epoch's losses have been read back, draws its noise from the global CPU generator exactly where the
reference does (compute_noise at :234), moves the generated batch to the host before touching numpy
(the reference's `images[k].data.numpy()` raises on device tensors), and
writes the sample grid as a PNG without torchvision / PIL (both absent here): same layout as
torchvision.utils.save_image (make_grid: padding 2, pad value 0; x*255+0.5 clamped to uint8)."""
import os
import struct
import zlib

import numpy as np
import torch


def make_grid(images, nrow, padding=2):
    """images [N, H, W] in [0, 1] (host) -> one [Hg, Wg] float array laid out like torchvision."""
    n, h, w = images.shape
    xmaps = min(nrow, n)
    ymaps = -(-n // xmaps)
    H, W = h + padding, w + padding
    grid = np.zeros((H * ymaps + padding, W * xmaps + padding), dtype=np.float32)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= n:
                break
            grid[y * H + padding:y * H + padding + h, x * W + padding:x * W + padding + w] = images[k]
            k += 1
    return grid


def write_png_gray(path, img01):
    """8-bit grayscale PNG of a [H, W] array in [0, 1]."""
    a = np.clip(img01 * 255.0 + 0.5, 0, 255).astype(np.uint8)
    h, w = a.shape
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


def generate_images(trainer, epoch, num_outputs=36, save=True, outdir=None):
    """ns_gan.py:228-262.  Returns the host array [num_outputs, shape, shape]."""
    m = trainer.model
    m.eval()
    with torch.no_grad():
        noise = trainer.compute_noise(num_outputs, m.z_dim)          # same CPU-generator draw as the reference
        images = m.G(noise)
    images = images.view(images.shape[0], m.shape, m.shape, -1).squeeze(-1).detach().float().cpu().numpy()
    grid_size = int(num_outputs ** 0.5)
    try:                                                             # the figure, when matplotlib is there
        import matplotlib.pyplot as plt
        from itertools import product
        plt.close()
        fig, ax = plt.subplots(grid_size, grid_size, figsize=(5, 5))
        for k, (i, j) in enumerate(product(range(grid_size), range(grid_size))):
            ax[i, j].get_xaxis().set_visible(False)
            ax[i, j].get_yaxis().set_visible(False)
            ax[i, j].imshow(images[k], cmap="gray")
    except Exception:                                                # noqa: BLE001  (headless / absent)
        pass
    if save:
        outname = os.path.join(outdir if outdir is not None else os.path.join("..", "viz"), trainer.name)
        os.makedirs(outname, exist_ok=True)
        write_png_gray(os.path.join(outname, "reconst_%d.png" % epoch), make_grid(images, grid_size))
    return images


def viz_loss(trainer):
    """ns_gan.py:264-281."""
    import matplotlib.pyplot as plt
    plt.style.use("ggplot")
    plt.rcParams["figure.figsize"] = (8, 6)
    xs = np.linspace(1, trainer.num_epochs, len(trainer.Dlosses))
    plt.plot(xs, trainer.Dlosses, "r")
    plt.plot(xs, trainer.Glosses, "g")
    plt.legend(["Discriminator", "Generator"])
    plt.title(trainer.name)
    plt.show()
