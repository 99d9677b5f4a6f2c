"""Pin the CPU oracle (oracle/sdmatte_oracle.py) to vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU-only."""
import os

import numpy as np
import torch

from oracle import sdmatte_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_g1_preprocess_matches_reference(golden_dir):
    g = _load(golden_dir, "g1_node_prepost.npz")
    image, tri = torch.from_numpy(g["image"]), torch.from_numpy(g["trimap"])
    d = O.preprocess(image, tri, int(g["inference_size"]), False)
    assert torch.equal(d["image"], torch.from_numpy(g["data_image"]))
    assert torch.equal(d["trimap"], torch.from_numpy(g["data_trimap"]))
    assert d["is_trans"].dtype == torch.int64
    assert torch.equal(d["is_trans"], torch.from_numpy(g["data_is_trans"]))
    assert torch.equal(d["trimap_coords"], torch.from_numpy(g["data_trimap_coords"]))
    assert len(d["caption"]) == int(g["data_caption_len"])
    d2 = O.preprocess(image, tri, int(g["inference_size"]), True)
    assert torch.equal(d2["is_trans"], torch.from_numpy(g["data_is_trans_transparent"]))


def test_g1_postprocess_matches_reference(golden_dir):
    g = _load(golden_dir, "g1_node_prepost.npz")
    image, tri = torch.from_numpy(g["image"]), torch.from_numpy(g["trimap"])
    fake = torch.from_numpy(g["fake_alpha"])
    for tag in g["cases"]:
        tag = str(tag)
        mode, refine, c = tag.split("__")
        refine = refine == "refine1"
        c = int(c[1:]) / 10.0
        a, m = O.postprocess(fake, image, tri, mode, refine, c)
        assert torch.equal(a, torch.from_numpy(g[f"alpha__{tag}"])), tag
        assert torch.equal(m, torch.from_numpy(g[f"matted__{tag}"])), tag


def test_g2_mask_pyramid(golden_dir):
    g = _load(golden_dir, "g2_mask_pyramid.npz")
    tri = torch.from_numpy(g["trimap_m11"])
    m = torch.nn.functional.interpolate((tri + 1) / 2, scale_factor=1 / 8, mode="nearest").flatten(start_dim=1)
    assert torch.equal(m, torch.from_numpy(g["attention_mask"]))
    # nearest x1/8 picks pixel (8i, 8j)
    assert torch.equal(m, ((tri + 1) / 2)[:, 0, ::8, ::8].flatten(1))
    bias = ((1 - m) * -10000.0).unsqueeze(1)
    assert torch.equal(bias, torch.from_numpy(g["bias_level0"]))
    l = int(round(bias.shape[-1] ** 0.5))
    for lev, heads in ((0, 5), (1, 10), (2, 20), (3, 20)):
        t = (l >> lev) ** 2
        pm = O.prepare_attention_mask(bias, t, heads)
        ref = torch.from_numpy(g[f"prepared_level{lev}_heads{heads}"])
        assert torch.equal(pm, ref)
        # closed form used by the HIP engine: level-k bias = bias_0[2^k i, 2^k j], image-major heads
        s = 1 << lev
        cf = bias.view(-1, l, l)[:, ::s, ::s].reshape(-1, 1, t).repeat_interleave(heads, dim=0)
        assert torch.equal(cf, ref)


def test_g3_attention_scores(golden_dir):
    g = _load(golden_dir, "g3_attention_scores.npz")
    q, k, v = (torch.from_numpy(g[n]) for n in ("q", "k", "v"))
    kb = torch.from_numpy(g["key_bias"])
    scale = float(g["scale"])
    pb = O.attention_scores(q, k, kb, scale)
    pn = O.attention_scores(q, k, None, scale)
    assert torch.allclose(pb, torch.from_numpy(g["probs_bias"]), rtol=0, atol=1e-7)
    assert torch.allclose(pn, torch.from_numpy(g["probs_nobias"]), rtol=0, atol=1e-7)
    # attention_core on the [B, L, h*d] layout reproduces bmm(probs, v) with image-major heads
    BH, Lq, d = q.shape
    heads = 2
    B = BH // heads
    tok = lambda x: x.view(B, heads, x.shape[1], d).permute(0, 2, 1, 3).reshape(B, x.shape[1], heads * d)
    o = O.attention_core(tok(q), tok(k), tok(v), heads, kb)
    ref = tok(torch.from_numpy(g["out_bias"]))
    assert torch.allclose(o, ref, rtol=0, atol=2e-6)


def test_g4_conv_in_surgery(golden_dir):
    g = _load(golden_dir, "g4_conv_in_surgery.npz")
    w0, b0 = torch.from_numpy(g["w0"]), torch.from_numpy(g["b0"])
    w8, b8 = O.conv_in_surgery(w0, b0, 2)
    assert int(g["in_channels"]) == 8
    assert torch.equal(w8, torch.from_numpy(g["conv_in_w"]))
    assert torch.equal(b8, torch.from_numpy(g["conv_in_b"]))
    aw, ab = O.aux_conv_in_init(w0, b0, 1024)
    assert torch.equal(aw, torch.from_numpy(g["aux_w"]))
    assert torch.equal(ab, torch.from_numpy(g["aux_b"]))


def test_timestep_embedding_probe_values():
    # SURVEY Appendix A.1 [probe]: dim=320, t=1 -> cos part 0.5403,0.5865,0.6284; sin part 0.8415,0.8099,0.7779
    e = O.get_timestep_embedding(torch.tensor([0.0, 1.0]), 320, True, 0.0)
    assert torch.allclose(e[0, :160], torch.ones(160)) and torch.allclose(e[0, 160:], torch.zeros(160))
    assert torch.allclose(e[1, :3], torch.tensor([0.5403, 0.5865, 0.6284]), atol=1e-4)
    assert torch.allclose(e[1, 160:163], torch.tensor([0.8415, 0.8099, 0.7779]), atol=1e-4)
