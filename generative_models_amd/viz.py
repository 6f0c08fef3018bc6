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


def generate_images(trainer, epoch, num_outputs=36, save=True, outdir=None, noise=None):
    """ns_gan.py:228-262.  Returns the host array [num_outputs, shape, shape].  noise: the generator
    input when the trainer's compute_noise takes more than (n, z_dim) (InfoGAN)."""
    m = trainer.model
    m.eval()
    with torch.no_grad():
        if noise is None:
            noise = trainer.compute_noise(num_outputs, m.z_dim)      # same CPU-generator draw as the reference
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


# ---- VAE family (vae.py:225-362, ae.py:166-205, bir_vae.py:234-374) -----------------------------
def _outdir(trainer, outdir):
    d = os.path.join(outdir if outdir is not None else os.path.join("..", "viz"), trainer.name)
    os.makedirs(d, exist_ok=True)
    return d


def _to_host_images(t, shape):
    return t.detach().float().cpu().reshape(t.shape[0], shape, shape).numpy()


def vae_sample_images(trainer, epoch=-100, num_images=36, save=True, outdir=None):
    """vae.py:254-276: z ~ N(0, I) from the global CPU generator (the reference's own draw, so the
    stream position after an epoch with viz on matches), decoded, written as ../viz/<name>/sample_<epoch>.png."""
    from .trainers import to_cuda
    m = trainer.model
    with torch.no_grad():
        z = to_cuda(torch.randn(num_images, m.z_dim))
        sample = m.decoder(z)
    images = _to_host_images(sample, m.shape)
    if save:
        write_png_gray(os.path.join(_outdir(trainer, outdir), "sample_%d.png" % epoch),
                       make_grid(images, int(num_images ** 0.5)))
    return images


def vae_reconstruct_images(trainer, images, epoch, save=True, outdir=None):
    """vae.py:225-252 / ae.py:166-193: the debugging batch through the model (a VAE samples eps here,
    as the reference does), real.png + reconst_<epoch>.png."""
    from .trainers import to_cuda
    m = trainer.model
    with torch.no_grad():
        batch = to_cuda(images.view(images.shape[0], -1))
        out = m(batch)
        out = out[0] if isinstance(out, tuple) else out
    side = int(round(out.shape[1] ** 0.5))
    rec = _to_host_images(out, side)
    if save:
        d = _outdir(trainer, outdir)
        grid = int(rec.shape[0] ** 0.5)
        write_png_gray(os.path.join(d, "real.png"), make_grid(_to_host_images(images.reshape(images.shape[0], -1), side), grid))
        write_png_gray(os.path.join(d, "reconst_%d.png" % epoch), make_grid(rec, grid))
    return rec


def vae_sample_interpolated_images(trainer):
    """vae.py:278-293: two latent vectors from p(z), decoded along the line between them (z_dim steps).
    Returns the list of decoded images (the reference displays them)."""
    from .trainers import to_cuda
    m = trainer.model
    z1 = torch.normal(torch.zeros(m.z_dim), 1)
    z2 = torch.normal(torch.zeros(m.z_dim), 1)
    out = []
    with torch.no_grad():
        for alpha in np.linspace(0, 1, m.z_dim):
            z = to_cuda((alpha * z1 + (1 - alpha) * z2).float().view(1, -1))
            out.append(_to_host_images(m.decoder(z), m.shape)[0])
    return out


def vae_viz_loss(trainer, second="kl_loss"):
    """vae.py:348-362: reconstruction loss in red, the second term (KL / MMD) in green."""
    import matplotlib.pyplot as plt
    plt.style.use("ggplot")
    plt.rcParams["figure.figsize"] = (8, 6)
    plt.plot(np.linspace(1, max(1, trainer.num_epochs), len(trainer.recon_loss)), trainer.recon_loss, "r")
    other = getattr(trainer, second, None)
    if other:
        plt.plot(np.linspace(1, max(1, trainer.num_epochs), len(other)), other, "g")
        plt.legend(["Reconstruction", "KL Divergence" if second == "kl_loss" else "MMD"])
    else:
        plt.legend(["Reconstruction"])
    plt.title(trainer.name)
    plt.show()
