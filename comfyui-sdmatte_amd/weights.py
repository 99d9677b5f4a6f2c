"""Checkpoint key schema of `SDMatte*.safetensors` and a deterministic synthetic-weight generator.

The schema follows the `nn.Module` attribute names of the reference model
(/root/reference/src/modeling/SDMatte/meta_arch.py:87-93 -> `vae.`, `unet.`, `text_encoder.`;
/root/reference/src/utils/replace.py:184-362 for the U-Net; /root/reference/src/utils/utils.py:33-41
for `unet.aux_conv_in`) and the diffusers block naming they instantiate.
`text_encoder.*` is dead on this path (replace.py:414-416 with meta_arch.py:48,77) and is neither
generated nor loaded.

There is no network and no real checkpoint in the build/GPU containers, so benchmarks and parity
tests run on synthetic weights: seed-fixed, conv/linear ~ N(0, 1/fan_in), norm gamma=1 beta=0,
biases small N(0, 0.02) (except that biases are non-zero so that bias handling is
actually exercised by the parity tests).
"""
from collections import OrderedDict

import torch

from .config import SDMatteConfig


def _resnet(sch, p, cin, cout, temb):
    sch[p + ".norm1.weight"] = (cin,)
    sch[p + ".norm1.bias"] = (cin,)
    sch[p + ".conv1.weight"] = (cout, cin, 3, 3)
    sch[p + ".conv1.bias"] = (cout,)
    if temb:
        sch[p + ".time_emb_proj.weight"] = (cout, temb)
        sch[p + ".time_emb_proj.bias"] = (cout,)
    sch[p + ".norm2.weight"] = (cout,)
    sch[p + ".norm2.bias"] = (cout,)
    sch[p + ".conv2.weight"] = (cout, cout, 3, 3)
    sch[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        sch[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1)
        sch[p + ".conv_shortcut.bias"] = (cout,)


def _vae_attn(sch, p, c):
    sch[p + ".group_norm.weight"] = (c,)
    sch[p + ".group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        sch[f"{p}.{n}.weight"] = (c, c)
        sch[f"{p}.{n}.bias"] = (c,)


def _transformer(sch, p, c, ctx):
    sch[p + ".norm.weight"] = (c,)
    sch[p + ".norm.bias"] = (c,)
    sch[p + ".proj_in.weight"] = (c, c)
    sch[p + ".proj_in.bias"] = (c,)
    b = p + ".transformer_blocks.0"
    for n in ("norm1", "norm2", "norm3"):
        sch[f"{b}.{n}.weight"] = (c,)
        sch[f"{b}.{n}.bias"] = (c,)
    for a, kdim in (("attn1", c), ("attn2", ctx)):
        sch[f"{b}.{a}.to_q.weight"] = (c, c)
        sch[f"{b}.{a}.to_k.weight"] = (c, kdim)
        sch[f"{b}.{a}.to_v.weight"] = (c, kdim)
        sch[f"{b}.{a}.to_out.0.weight"] = (c, c)
        sch[f"{b}.{a}.to_out.0.bias"] = (c,)
    sch[b + ".ff.net.0.proj.weight"] = (8 * c, c)
    sch[b + ".ff.net.0.proj.bias"] = (8 * c,)
    sch[b + ".ff.net.2.weight"] = (c, 4 * c)
    sch[b + ".ff.net.2.bias"] = (c,)
    sch[p + ".proj_out.weight"] = (c, c)
    sch[p + ".proj_out.bias"] = (c,)


def unet_up_resnet_in_channels(cfg: SDMatteConfig):
    """Input channels of the 3 resnets of each up block (diffusers get_up_block bookkeeping,
    driven by replace.py:295-349).  Full model: (2560,2560,2560),(2560,2560,1920),(1920,1280,960),(960,640,640)."""
    ch = cfg.unet_channels
    rev = list(reversed(ch))
    out = []
    output_channel = rev[0]
    n = cfg.unet_layers_per_block + 1
    for i in range(len(ch)):
        prev_output_channel = output_channel
        output_channel = rev[i]
        input_channel = rev[min(i + 1, len(ch) - 1)]
        ins = []
        for j in range(n):
            res_skip = input_channel if j == n - 1 else output_channel
            resnet_in = prev_output_channel if j == 0 else output_channel
            ins.append(resnet_in + res_skip)
        out.append(tuple(ins))
    return out


def weight_schema(cfg: SDMatteConfig) -> "OrderedDict[str, tuple]":
    sch = OrderedDict()
    # ---------------- VAE encoder ----------------
    vc = cfg.vae_channels
    sch["vae.encoder.conv_in.weight"] = (vc[0], 3, 3, 3)
    sch["vae.encoder.conv_in.bias"] = (vc[0],)
    cprev = vc[0]
    for i, c in enumerate(vc):
        for j in range(cfg.vae_layers_per_block):
            _resnet(sch, f"vae.encoder.down_blocks.{i}.resnets.{j}", cprev if j == 0 else c, c, 0)
        cprev = c
        if i < len(vc) - 1:
            sch[f"vae.encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3)
            sch[f"vae.encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
    cm = vc[-1]
    _resnet(sch, "vae.encoder.mid_block.resnets.0", cm, cm, 0)
    _vae_attn(sch, "vae.encoder.mid_block.attentions.0", cm)
    _resnet(sch, "vae.encoder.mid_block.resnets.1", cm, cm, 0)
    sch["vae.encoder.conv_norm_out.weight"] = (cm,)
    sch["vae.encoder.conv_norm_out.bias"] = (cm,)
    lc = cfg.vae_latent_channels
    sch["vae.encoder.conv_out.weight"] = (2 * lc, cm, 3, 3)
    sch["vae.encoder.conv_out.bias"] = (2 * lc,)
    sch["vae.quant_conv.weight"] = (2 * lc, 2 * lc, 1, 1)
    sch["vae.quant_conv.bias"] = (2 * lc,)
    sch["vae.post_quant_conv.weight"] = (lc, lc, 1, 1)
    sch["vae.post_quant_conv.bias"] = (lc,)
    # ---------------- VAE decoder ----------------
    sch["vae.decoder.conv_in.weight"] = (cm, lc, 3, 3)
    sch["vae.decoder.conv_in.bias"] = (cm,)
    _resnet(sch, "vae.decoder.mid_block.resnets.0", cm, cm, 0)
    _vae_attn(sch, "vae.decoder.mid_block.attentions.0", cm)
    _resnet(sch, "vae.decoder.mid_block.resnets.1", cm, cm, 0)
    rev = list(reversed(vc))
    cprev = rev[0]
    for i, c in enumerate(rev):
        for j in range(cfg.vae_layers_per_block + 1):
            _resnet(sch, f"vae.decoder.up_blocks.{i}.resnets.{j}", cprev if j == 0 else c, c, 0)
        cprev = c
        if i < len(rev) - 1:
            sch[f"vae.decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
            sch[f"vae.decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    sch["vae.decoder.conv_norm_out.weight"] = (rev[-1],)
    sch["vae.decoder.conv_norm_out.bias"] = (rev[-1],)
    sch["vae.decoder.conv_out.weight"] = (3, rev[-1], 3, 3)
    sch["vae.decoder.conv_out.bias"] = (3,)
    # ---------------- U-Net ----------------
    uc = cfg.unet_channels
    te = cfg.time_embed_dim
    ctx = cfg.cross_attention_dim
    sch["unet.conv_in.weight"] = (uc[0], cfg.unet_in_channels, 3, 3)
    sch["unet.conv_in.bias"] = (uc[0],)
    sch["unet.aux_conv_in.weight"] = (ctx, 4, 3, 3)
    sch["unet.aux_conv_in.bias"] = (ctx,)
    for name, din in (("time_embedding", uc[0]), ("point_embedding", cfg.point_embeddings_input_dim),
                      ("bbox_embedding", cfg.bbox_embeddings_input_dim)):
        sch[f"unet.{name}.linear_1.weight"] = (te, din)
        sch[f"unet.{name}.linear_1.bias"] = (te,)
        sch[f"unet.{name}.linear_2.weight"] = (te, te)
        sch[f"unet.{name}.linear_2.bias"] = (te,)
    cprev = uc[0]
    nlev = len(uc)
    for i, c in enumerate(uc):
        has_attn = i < nlev - 1
        for j in range(cfg.unet_layers_per_block):
            _resnet(sch, f"unet.down_blocks.{i}.resnets.{j}", cprev if j == 0 else c, c, te)
            if has_attn:
                _transformer(sch, f"unet.down_blocks.{i}.attentions.{j}", c, ctx)
        cprev = c
        if i < nlev - 1:
            sch[f"unet.down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3)
            sch[f"unet.down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
    cm = uc[-1]
    _resnet(sch, "unet.mid_block.resnets.0", cm, cm, te)
    _transformer(sch, "unet.mid_block.attentions.0", cm, ctx)
    _resnet(sch, "unet.mid_block.resnets.1", cm, cm, te)
    rev = list(reversed(uc))
    ins = unet_up_resnet_in_channels(cfg)
    for i, c in enumerate(rev):
        has_attn = i > 0
        for j in range(cfg.unet_layers_per_block + 1):
            _resnet(sch, f"unet.up_blocks.{i}.resnets.{j}", ins[i][j], c, te)
            if has_attn:
                _transformer(sch, f"unet.up_blocks.{i}.attentions.{j}", c, ctx)
        if i < nlev - 1:
            sch[f"unet.up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
            sch[f"unet.up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    sch["unet.conv_norm_out.weight"] = (uc[0],)
    sch["unet.conv_norm_out.bias"] = (uc[0],)
    sch["unet.conv_out.weight"] = (cfg.unet_out_channels, uc[0], 3, 3)
    sch["unet.conv_out.bias"] = (cfg.unet_out_channels,)
    return sch


def count_params(cfg: SDMatteConfig, prefix: str = "") -> int:
    n = 0
    for k, s in weight_schema(cfg).items():
        if k.startswith(prefix):
            m = 1
            for d in s:
                m *= d
            n += m
    return n


def _is_norm(key: str) -> bool:
    parts = key.split(".")
    return any(p.startswith("norm") or p in ("group_norm", "conv_norm_out") for p in parts[:-1])


def synthetic_state_dict(cfg: SDMatteConfig, seed: int = 0, gain: float = 1.0):
    """Deterministic fp32 CPU state dict with the exact checkpoint key schema."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shape in weight_schema(cfg).items():
        if _is_norm(k):
            if k.endswith(".weight"):
                t = 1.0 + 0.1 * torch.randn(shape, generator=g)
            else:
                t = 0.05 * torch.randn(shape, generator=g)
        elif k.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) * (gain / fan_in ** 0.5)
        sd[k] = t.contiguous()
    return sd
