"""Image-tile partition across ranks and reassembly of the gathered framebuffer.

Replaces the column-per-job pool and mpsc gather of the reference's render()
(src/world.rs:38-74): tiles of T x T pixels are numbered row-major and dealt round-robin,
tile i -> rank i % nranks (interleaving balances background tiles against deep-bounce
tiles).  Every rank produces its tiles packed [local tile][row][col][rgb]; one all-gather
(RCCL over xGMI, or gloo in the CPU tests) collects them and `assemble` de-interleaves.
The same packing is what rtx_render_device() writes (include/rtx.h).
"""
import torch


def tiles_xy(width, height, tile):
    return (width + tile - 1) // tile, (height + tile - 1) // tile


def local_tile_count(width, height, tile, rank, nranks):
    tx, ty = tiles_xy(width, height, tile)
    nt = tx * ty
    return (nt + nranks - 1 - rank) // nranks


def padded_tiles_per_rank(width, height, tile, nranks):
    """All ranks gather equal-sized buffers: the count of rank 0 (the largest)."""
    return local_tile_count(width, height, tile, 0, nranks)


def pack_region(image, tile, rank, nranks):
    """Packs this rank's tiles of a full H x W x 3 image tensor (used by the CPU tests, where the
    oracle stands in for the device renderer)."""
    h, w = image.shape[0], image.shape[1]
    tx, ty = tiles_xy(w, h, tile)
    n = padded_tiles_per_rank(w, h, tile, nranks)
    out = torch.zeros((n, tile, tile, 3), dtype=image.dtype)
    for lt in range(local_tile_count(w, h, tile, rank, nranks)):
        gt = lt * nranks + rank
        x0, y0 = (gt % tx) * tile, (gt // tx) * tile
        patch = image[y0:y0 + tile, x0:x0 + tile]
        out[lt, :patch.shape[0], :patch.shape[1]] = patch
    return out


def assemble(gathered, width, height, tile, nranks):
    """gathered: [nranks, tiles_per_rank, T, T, 3] (any device) -> [H, W, 3]."""
    tx, ty = tiles_xy(width, height, tile)
    n = gathered.shape[1]
    # global tile index = local * nranks + rank
    t = gathered.permute(1, 0, 2, 3, 4).reshape(n * nranks, tile, tile, 3)[: tx * ty]
    img = t.reshape(ty, tx, tile, tile, 3).permute(0, 2, 1, 3, 4).reshape(ty * tile, tx * tile, 3)
    return img[:height, :width].contiguous()


def gather_image(local_tiles, width, height, tile, rank, nranks, dist=None):
    """All-gathers every rank's packed tiles and returns the assembled image (on every rank)."""
    n = padded_tiles_per_rank(width, height, tile, nranks)
    flat = local_tiles.reshape(-1)
    need = n * tile * tile * 3
    if flat.numel() < need:
        flat = torch.cat([flat, torch.zeros(need - flat.numel(), dtype=flat.dtype, device=flat.device)])
    if nranks == 1:
        return assemble(flat.reshape(1, n, tile, tile, 3), width, height, tile, 1)
    dev = flat.device
    if dist.get_backend() != "nccl" and dev.type != "cpu":  # rehearsal backends gather through the host
        flat = flat.cpu()
    out = torch.empty(nranks * need, dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, flat.contiguous())
    return assemble(out.reshape(nranks, n, tile, tile, 3).to(dev), width, height, tile, nranks)
