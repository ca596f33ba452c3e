"""Row-band sharding of one canvas across the GPUs of a node.

Rows never interact in this path (the prefix sum runs along a scanline, S:983; compositing is per
pixel), so rank r of N simply renders the bands {r, r+N, r+2N, ...} of the viewport (band =
``tile_rows`` scanlines; strips of a few bands are interleaved for load balance) from the SAME scene description.  An edge that
crosses a band border is handed to both owners: duplicated edges are the "halo", no pixel ever
crosses a GPU, and the data path needs no collective.  The only communication is the optional
assembly of the finished bands (all_gather over RCCL on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

import numpy as np


def n_bands(rows: int, tile_rows: int) -> int:
    return (rows + tile_rows - 1) // tile_rows


def default_strip_bands(rows: int, tile_rows: int, world: int) -> int:
    """Bands per strip when the caller does not choose: ONE strip per rank -- a rank flattens every path that reaches one of its
    strips completely, so the tallest strips duplicate the least geometry (round 4, emulated on one GPU, config 4: the slowest of
    8 / 4 / 2 ranks 0.1249 -> 0.1225, 0.1972 -> 0.1925, 0.3551 -> 0.3489 ms against two interleaved strips per rank) --, not
    under 128 scanlines where the canvas allows it, and never so tall that a rank is left without rows.  A drawing whose content is bunched in a few rows wants `SVGR_STRIP_BANDS` smaller."""
    nb, world = n_bands(rows, tile_rows), max(world, 1)
    strip = -(-nb // world)                       # one strip per rank ...
    if nb >= world and -(-nb // strip) < world:   # ... unless rounding up leaves a rank without rows (17 bands on 8 ranks: strips of 3
        strip = nb // world                       #     are 6 strips): then the shorter strip, and some ranks own two
    floor_ = max(1, 128 // tile_rows)             # (not under 128 scanlines -- as long as that does not idle a rank either)
    if strip < floor_ and -(-nb // floor_) >= min(world, nb):
        strip = floor_
    return max(strip, 1)


def owned_bands(rows: int, tile_rows: int, rank: int, world: int, strip: int = 1) -> list[int]:
    """Bands of rank `rank`: the strips s (of `strip` consecutive bands) with s % world == rank."""
    return [b for b in range(n_bands(rows, tile_rows)) if (b // strip) % world == rank]


def owned_row_ranges(rows: int, tile_rows: int, rank: int, world: int, strip: int = 1) -> list[tuple[int, int]]:
    """[(row0, row1)) of every owned band, in the order they are packed in the rank's buffer."""
    return [(b * tile_rows, min((b + 1) * tile_rows, rows)) for b in owned_bands(rows, tile_rows, rank, world, strip)]


def max_owned_bands(rows: int, tile_rows: int, world: int, strip: int = 1) -> int:
    return max(len(owned_bands(rows, tile_rows, r, world, strip)) for r in range(world))


def assemble(parts, rows: int, tile_rows: int, strip: int = 1):
    """parts[r] = rank r's packed bands, shape (k_r * tile_rows [or more, padded], cols, ch) -> full
    (rows, cols, ch) canvas.  Works on numpy arrays and torch tensors alike."""
    world = len(parts)
    first = parts[0]
    out = first.new_zeros((rows,) + tuple(first.shape[1:])) if hasattr(first, "new_zeros") else np.zeros(
        (rows,) + tuple(first.shape[1:]), dtype=first.dtype)
    for r, part in enumerate(parts):
        for k, (r0, r1) in enumerate(owned_row_ranges(rows, tile_rows, r, world, strip)):
            out[r0:r1] = part[k * tile_rows: k * tile_rows + (r1 - r0)]
    return out


def gather_canvas(local, rows: int, tile_rows: int, group=None, strip: int = 1):
    """all_gather the ranks' packed bands (torch tensor, CPU/gloo or GPU/RCCL) into the full canvas on every rank.

    Strips are dealt round-robin, so `world` consecutive strips -- one of each rank, in rank order -- are a contiguous run of
    canvas rows: exactly the layout all_gather_into_tensor produces.  Every complete round of strips is therefore gathered
    STRAIGHT into its rows of the final canvas (one collective per round: one with the default of a strip per rank), no
    staging buffer and no assembly copy; only a ragged tail (a last round in which some rank has a short strip or none) takes
    the padded gather + copy, for its rows alone."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    srows = strip * tile_rows                      # scanlines per strip
    rounds = rows // (srows * world)               # complete rounds of strips
    full = torch.empty((rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    local = local.contiguous()
    for g in range(rounds):
        dist.all_gather_into_tensor(full[g * world * srows:(g + 1) * world * srows], local[g * srows:(g + 1) * srows], group=group)
    tail0 = rounds * world * srows
    if tail0 < rows:
        # the rest: rank r's share is what it has beyond the complete rounds (possibly nothing), padded to the largest share
        rank = dist.get_rank(group)
        shares = [min(max(rows - tail0 - r * srows, 0), srows) for r in range(world)]
        pad_rows = max(shares)
        mine = local[rounds * srows: rounds * srows + shares[rank]]
        if mine.shape[0] < pad_rows:
            mine = torch.cat([mine, torch.zeros((pad_rows - mine.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                                                device=local.device)], dim=0)
        got = torch.empty((world * pad_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(got, mine.contiguous(), group=group)
        at = tail0
        for r in range(world):
            full[at:at + shares[r]] = got[r * pad_rows: r * pad_rows + shares[r]]
            at += shares[r]
    return full


def gather_canvas_staged(local, rows: int, tile_rows: int, group=None, strip: int = 1):
    """The same result through one padded all_gather and an assembly copy (what gather_canvas replaced; kept as its check)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    pad_rows = max_owned_bands(rows, tile_rows, world, strip) * tile_rows
    if local.shape[0] < pad_rows:
        pad = torch.zeros((pad_rows - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    # concatenated layout (world * pad_rows, ...): accepted by both the gloo and the RCCL back-ends
    gathered = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, local.contiguous(), group=group)
    gathered = gathered.view((world,) + tuple(local.shape))
    return assemble([gathered[r] for r in range(world)], rows, tile_rows, strip)
