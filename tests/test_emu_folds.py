"""fold == unfold: every exact algebraic fold the engine applies at load time (SURVEY.md 8a "Exact algebraic folds", each "needs a
fold == unfold CPU test") against the unfolded torch computation of the reference graph, on the kernel emulator:
  (i)   time / opacity / bbox embedding constants folded into each ResBlock's conv1 bias table (replace.py:419-459 + A.3);
  (ii)  aux_conv_in folded into every cross-attention K|V projection (meta_arch.py:215-218, utils.py:33-41);
  (iii) the attention logit scale d^-1/2 * log2(e) folded into the to_q weights (replace.py:75-122)."""
import ctypes
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _engine(cfg, w, precision):
    from emu.build_emu import build
    from comfyui_sdmatte_amd.engine import Bindings, Engine
    eng = Engine(cfg, 0, True, _lib=Bindings(ctypes.CDLL(build())), precision=precision)
    eng.load_state_dict(w)
    return eng


def _setup(seed=2):
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.weights import synthetic_state_dict
    cfg = SDMatteConfig.tiny()
    w = synthetic_state_dict(cfg, seed)
    # synthetic biases are zero: give the folded ones something to fold
    g = torch.Generator().manual_seed(5)
    for k in list(w):
        if k.endswith("aux_conv_in.bias") or k.endswith("conv1.bias") or k.endswith("time_emb_proj.bias"):
            w[k] = torch.randn(w[k].shape, generator=g) * 0.1
    return cfg, w


def test_cross_attention_kv_fold_equals_aux_conv_then_projections(pkg):
    cfg, w = _setup()
    g = torch.Generator().manual_seed(11)
    z = torch.randn(2, 4, 8, 8, generator=g)                          # trimap latent
    ctx = F.conv2d(z, w["unet.aux_conv_in.weight"], w["unet.aux_conv_in.bias"], padding=1)       # meta_arch.py:216
    tokens = ctx.permute(0, 2, 3, 1)                                  # [B,l,l,ctx] (= .view(B,ctx,L).permute(0,2,1), :217-218)
    x16 = torch.zeros(2, 8, 8, 16)
    x16[..., 4:8] = z.permute(0, 2, 3, 1)                             # the latent sits at channels 4..7 of the U-Net input tensor
    blocks = ["unet.down_blocks.0.attentions.0", "unet.mid_block.attentions.0", "unet.up_blocks.3.attentions.2"]
    for precision, rtol in (("fp16x3", 2e-5), ("fp16", 4e-3)):
        eng = _engine(cfg, w, precision)
        for b in blocks:
            p = b + ".transformer_blocks.0.attn2"
            C = w[p + ".to_k.weight"].shape[0]
            want = torch.cat([F.linear(tokens, w[p + ".to_k.weight"]), F.linear(tokens, w[p + ".to_v.weight"])], dim=-1)
            got = eng.debug_run_layer(p + ".kv_folded", x16, 2 * C)
            err = (got - want).abs().max().item() / want.abs().max().item()
            assert got.shape == want.shape and err < rtol, (precision, b, err)
        eng.close()


def test_logit_scale_fold_into_to_q(pkg):
    cfg, w = _setup()
    g = torch.Generator().manual_seed(12)
    p = "unet.down_blocks.1.attentions.0.transformer_blocks.0.attn1"
    C = w[p + ".to_q.weight"].shape[0]
    x = torch.randn(1, 4, 4, C, generator=g)
    scale = (64 ** -0.5) * math.log2(math.e)                          # softmax(q.k * d^-1/2) evaluated as 2^(q'.k - max)
    want = torch.cat([F.linear(x, w[p + ".to_q.weight"]) * scale, F.linear(x, w[p + ".to_k.weight"]), F.linear(x, w[p + ".to_v.weight"])], dim=-1)
    eng = _engine(cfg, w, "fp16x3")
    got = eng.debug_run_layer(p + ".qkv", x, 3 * C)
    assert (got - want).abs().max().item() < 2e-5 * want.abs().max().item()
    eng.close()


def test_embedding_constants_fold_into_conv1_bias_tables(pkg):
    from oracle import sdmatte_oracle as O
    cfg, w = _setup()
    cd = cfg.as_dict()
    eng = _engine(cfg, w, "fp16")
    names = [f"unet.down_blocks.{i}.resnets.{j}" for i in range(4) for j in range(2)] + ["unet.mid_block.resnets.0", "unet.mid_block.resnets.1"] + \
            [f"unet.up_blocks.{i}.resnets.{j}" for i in range(4) for j in range(3)]
    for is_trans, coords in ((0, [0.0, 0.0, 1.0, 1.0]), (1, [0.1, 0.25, 0.7, 0.9])):
        trans = torch.tensor([1 - is_trans])                          # meta_arch.py:237-238
        coor = O.get_timestep_embedding(torch.tensor(coords), cfg.bbox_embeddings_input_dim // 4, True, 0.0)     # meta_arch.py:181-186
        emb = O.unet_embedding(w, cd, trans, coor)                    # replace.py:419-459
        for idx, n in enumerate(names):
            want = w[n + ".conv1.bias"] + F.linear(F.silu(emb), w[n + ".time_emb_proj.weight"], w[n + ".time_emb_proj.bias"])[0]
            got = eng.debug_temb_row(idx, is_trans, coords, want.numel())
            assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item()), (n, is_trans)
    eng.close()
