"""Synthetic image + three-valued trimap batches: seed-fixed U[0,1) RGB and a
disc(1.0)/annulus(0.5)/background(0.0) trimap per image.  Used by bench.py and the parity tests;
there is no dataset (and no network) on the GPU box."""
import torch


def synthetic_inputs(B: int, H: int, W: int, seed: int = 1234):
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(B, H, W, 3, generator=g)
    delta = (torch.rand(B, 2, generator=g) - 0.5) * 0.1
    yy = torch.arange(H, dtype=torch.float32)[None, :, None]
    xx = torch.arange(W, dtype=torch.float32)[None, None, :]
    cy = (0.5 + delta[:, 0])[:, None, None] * H
    cx = (0.5 + delta[:, 1])[:, None, None] * W
    s = float(min(H, W))
    r = torch.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
    trimap = torch.zeros(B, H, W)
    trimap[r < 0.40 * s] = 0.5
    trimap[r < 0.30 * s] = 1.0
    return image, trimap
