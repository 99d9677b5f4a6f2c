"""CPU ORACLE (test infrastructure, NOT product code) for the SDMatte `Apply SDMatte` hot path.

This file is a plain fp32 `torch` CPU restatement of the reference algorithm.  It exists only so
that tests/, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg can check / time the
hand-written HIP engine against it.  Nothing under `comfyui-sdmatte_amd/` imports it.

PARITY STATUS: *partially pinned*.
  * Pinned against outputs of the reference itself (golden vectors under tests/golden/, produced by
    tests/golden/make_golden.py importing /root/reference in the build container):
      - node pre/post-processing  (sdmatte_nodes.py:204-214,339-397)           -> G1
      - attention-mask pyramid     (replace.py:20-72, meta_arch.py:200-204)      -> G2
      - masked attention scores    (replace.py:75-122)                           -> G3
      - conv_in / aux_conv_in surgery (utils.py:13-41)                           -> G4
  * "parity unpinned" for everything that executes inside the un-vendored third-party dependency
    `diffusers` (requirements.txt:1 pins only `diffusers>=0.25.0`; not installed here, no network):
    AutoencoderKL Encoder/Decoder, ResnetBlock2D, Transformer2DModel/BasicTransformerBlock/Attention,
    Down/Upsample2D, Timesteps/TimestepEmbedding.  Those are restated from the library's published
    behaviour on top of stock torch primitives; the composition is anchored on
    the reference's own call sites (cited per function below) and on the exact reproduction of the
    SD-2.1 parameter counts by the key schema (tests/test_schema.py).  The reference has no tests.
  * "parity unpinned" as well for the prompt-type routing of `SDMatte.forward` (bbox_mask / mask / auto_mask /
    point_mask, point-coordinate padding, use_coor_input, attn_mask_aux_input): `meta_arch.py` cannot be imported
    (hard-coded `.cuda()`, diffusers), so `sdmatte_forward(..., aux_input=...)` restates meta_arch.py:131-206 and
    replace.py:446-457 line by line (cited in place) without a golden vector.
  * EXTENSION beyond the reference (no parity claim possible): rectangular token grids in `prepare_attention_mask`
    (`src_hw` / `dst_hw`); the reference asserts perfect squares (replace.py:59-60).  Square inputs take the reference branch.

The reference's `force_cpu=True` branch is the semantics restated here: fp32, no autocast, default
(un-sliced) AttnProcessor (sdmatte_nodes.py:355-360, utils.py:46).
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------------
def get_timestep_embedding(t: Tensor, dim: int, flip_sin_to_cos: bool = True,
                           downscale_freq_shift: float = 0.0, scale: float = 1.0,
                           max_period: int = 10000) -> Tensor:
    """[3P diffusers.models.embeddings.get_timestep_embedding]; call sites meta_arch.py:181-186 and
    `Timesteps` at replace.py:188.  SURVEY Appendix A.1."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32)
    exponent = exponent / (half - downscale_freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def timestep_embedding_mlp(w: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    """[3P TimestepEmbedding] linear_1 -> SiLU -> linear_2 (replace.py:190-200). Appendix A.2."""
    x = F.linear(x, w[p + ".linear_1.weight"], w[p + ".linear_1.bias"])
    x = F.silu(x)
    return F.linear(x, w[p + ".linear_2.weight"], w[p + ".linear_2.bias"])


def resnet_block(w, p: str, x: Tensor, temb: Optional[Tensor], groups: int, eps: float) -> Tensor:
    """[3P ResnetBlock2D] (Appendix A.3); reached from replace.py:476,495,519 and the VAE."""
    h = F.group_norm(x, groups, w[p + ".norm1.weight"], w[p + ".norm1.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, w[p + ".conv1.weight"], w[p + ".conv1.bias"], padding=1)
    if temb is not None:
        t = F.linear(F.silu(temb), w[p + ".time_emb_proj.weight"], w[p + ".time_emb_proj.bias"])
        h = h + t[:, :, None, None]
    h = F.group_norm(h, groups, w[p + ".norm2.weight"], w[p + ".norm2.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, w[p + ".conv2.weight"], w[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in w:
        x = F.conv2d(x, w[p + ".conv_shortcut.weight"], w[p + ".conv_shortcut.bias"])
    return x + h


def vae_attention(w, p: str, x: Tensor, groups: int, eps: float) -> Tensor:
    """[3P Attention, 1 head, residual_connection=True, SDPA] VAE mid-block attention (Appendix A.5).
    Not touched by replace_attention_mask_method (utils.py:44-56 walks the U-Net only)."""
    B, C, H, W = x.shape
    r = x
    h = x.view(B, C, H * W)
    h = F.group_norm(h, groups, w[p + ".group_norm.weight"], w[p + ".group_norm.bias"], eps)
    h = h.transpose(1, 2)                                       # [B, HW, C]
    q = F.linear(h, w[p + ".to_q.weight"], w[p + ".to_q.bias"])
    k = F.linear(h, w[p + ".to_k.weight"], w[p + ".to_k.bias"])
    v = F.linear(h, w[p + ".to_v.weight"], w[p + ".to_v.bias"])
    o = attention_core(q, k, v, heads=1, bias=None)
    o = F.linear(o, w[p + ".to_out.0.weight"], w[p + ".to_out.0.bias"])
    o = o.transpose(1, 2).reshape(B, C, H, W)
    return o + r


def prepare_attention_mask(bias_b1l: Tensor, target_length: int, heads: int, src_hw=None, dst_hw=None) -> Tensor:
    """Restatement of custom_prepare_attention_mask (replace.py:20-72): nearest-resize the
    additive bias, viewed as a square image, to the level's token grid; repeat per head.
    `src_hw` / `dst_hw` (EXTENSION, not in the reference, which asserts perfect squares at replace.py:59-60): the same
    nearest resize on a rectangular token grid."""
    B = bias_b1l.shape[0]
    cur = bias_b1l.shape[-1]
    if cur != target_length:
        if src_hw is not None and src_hw[0] != src_hw[1]:
            assert src_hw[0] * src_hw[1] == cur and dst_hw[0] * dst_hw[1] == target_length
            m = bias_b1l.view(B, -1, src_hw[0], src_hw[1])
            m = F.interpolate(m, size=tuple(dst_hw), mode="nearest")
        else:
            cs = int(math.sqrt(cur))
            ts = int(math.sqrt(target_length))
            assert cs * cs == cur and ts * ts == target_length        # replace.py:59-60
            m = bias_b1l.view(B, -1, cs, cs)
            m = F.interpolate(m, size=(ts, ts), mode="nearest")        # replace.py:62
        bias_b1l = m.view(B, 1, target_length)
    return bias_b1l.repeat_interleave(heads, dim=0)                # replace.py:65-67


def attention_scores(q: Tensor, k: Tensor, bias: Optional[Tensor], scale: float) -> Tensor:
    """Restatement of custom_get_attention_scores (replace.py:75-122) in fp32:
    softmax(baddbmm(bias, q, k^T, beta=1, alpha=scale))."""
    if bias is not None:
        s = torch.baddbmm(bias.expand(q.shape[0], q.shape[1], k.shape[1]), q, k.transpose(-1, -2),
                          beta=1, alpha=scale)
    else:
        s = torch.bmm(q, k.transpose(-1, -2)) * scale
    return s.softmax(dim=-1)


def attention_core(q: Tensor, k: Tensor, v: Tensor, heads: int, bias: Optional[Tensor]) -> Tensor:
    """[3P AttnProcessor.__call__ + head_to_batch_dim/batch_to_head_dim] (Appendix A.7).
    q [B,Lq,h*d], k/v [B,Lk,h*d], bias [B*h,1,Lk] or None.  One (image,head) at a time, which is
    numerically identical to the un-sliced bmm (the reference's own CUDA path uses
    SlicedAttnProcessor(slice_size=1), sdmatte_nodes.py:331-337) and keeps the score matrix small."""
    B, Lq, C = q.shape
    Lk = k.shape[1]
    d = C // heads
    scale = d ** -0.5
    qh = q.view(B, Lq, heads, d).permute(0, 2, 1, 3).reshape(B * heads, Lq, d)
    kh = k.view(B, Lk, heads, d).permute(0, 2, 1, 3).reshape(B * heads, Lk, d)
    vh = v.view(B, Lk, heads, d).permute(0, 2, 1, 3).reshape(B * heads, Lk, d)
    out = torch.empty_like(qh)
    # ... and, beyond 1024x1024 inputs, one block of query rows at a time (the softmax is per row: the same numbers), so that the score
    # matrix of a 65536-token level (2048x2048 input: 17 GB per image and head) never exists in full
    rows = max(1, min(Lq, (1 << 28) // max(Lk, 1)))
    for i in range(B * heads):
        b = None if bias is None else bias[i:i + 1]
        for r0 in range(0, Lq, rows):
            p = attention_scores(qh[i:i + 1, r0:r0 + rows], kh[i:i + 1], b, scale)
            out[i:i + 1, r0:r0 + rows] = torch.bmm(p, vh[i:i + 1])
    return out.view(B, heads, Lq, d).permute(0, 2, 1, 3).reshape(B, Lq, C)


def transformer_2d(w, p: str, x: Tensor, ehs: Tensor, bias_b1l: Optional[Tensor], heads: int,
                   groups: int, gn_eps: float, ln_eps: float, hw0=None) -> Tensor:
    """[3P Transformer2DModel(use_linear_projection=True) + BasicTransformerBlock] (Appendix A.7);
    attention internals per replace.py:20-122."""
    B, C, H, W = x.shape
    r = x
    h = F.group_norm(x, groups, w[p + ".norm.weight"], w[p + ".norm.bias"], gn_eps)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    h = F.linear(h, w[p + ".proj_in.weight"], w[p + ".proj_in.bias"])
    b = p + ".transformer_blocks.0"
    # self-attention with the trimap key bias
    n = F.layer_norm(h, (C,), w[b + ".norm1.weight"], w[b + ".norm1.bias"], ln_eps)
    q = F.linear(n, w[b + ".attn1.to_q.weight"])
    k = F.linear(n, w[b + ".attn1.to_k.weight"])
    v = F.linear(n, w[b + ".attn1.to_v.weight"])
    mb = None if bias_b1l is None else prepare_attention_mask(bias_b1l, H * W, heads, hw0, (H, W))
    a = attention_core(q, k, v, heads, mb)
    a = F.linear(a, w[b + ".attn1.to_out.0.weight"], w[b + ".attn1.to_out.0.bias"])
    h = h + a
    # cross-attention to the trimap-latent tokens (no mask: encoder_attention_mask is None)
    n = F.layer_norm(h, (C,), w[b + ".norm2.weight"], w[b + ".norm2.bias"], ln_eps)
    q = F.linear(n, w[b + ".attn2.to_q.weight"])
    k = F.linear(ehs, w[b + ".attn2.to_k.weight"])
    v = F.linear(ehs, w[b + ".attn2.to_v.weight"])
    a = attention_core(q, k, v, heads, None)
    a = F.linear(a, w[b + ".attn2.to_out.0.weight"], w[b + ".attn2.to_out.0.bias"])
    h = h + a
    # GEGLU feed-forward
    n = F.layer_norm(h, (C,), w[b + ".norm3.weight"], w[b + ".norm3.bias"], ln_eps)
    f = F.linear(n, w[b + ".ff.net.0.proj.weight"], w[b + ".ff.net.0.proj.bias"])
    u, g = f.chunk(2, dim=-1)
    f = u * F.gelu(g)
    f = F.linear(f, w[b + ".ff.net.2.weight"], w[b + ".ff.net.2.bias"])
    h = h + f
    h = F.linear(h, w[p + ".proj_out.weight"], w[p + ".proj_out.bias"])
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return h + r


# ----------------------------------------------------------------------------------------------
# VAE
# ----------------------------------------------------------------------------------------------
def vae_encoder(w, cfg: dict, x: Tensor) -> Tensor:
    """[3P AutoencoderKL.encoder] (Appendix A.5); call sites meta_arch.py:142,209."""
    g, eps = cfg["vae_groups"], cfg["vae_eps"]
    vc = cfg["vae_channels"]
    p = "vae.encoder"
    h = F.conv2d(x, w[p + ".conv_in.weight"], w[p + ".conv_in.bias"], padding=1)
    for i in range(len(vc)):
        for j in range(cfg["vae_layers_per_block"]):
            h = resnet_block(w, f"{p}.down_blocks.{i}.resnets.{j}", h, None, g, eps)
        if i < len(vc) - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)       # Appendix A.4 (VAE Downsample2D)
            h = F.conv2d(h, w[f"{p}.down_blocks.{i}.downsamplers.0.conv.weight"],
                         w[f"{p}.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2, padding=0)
    h = resnet_block(w, p + ".mid_block.resnets.0", h, None, g, eps)
    h = vae_attention(w, p + ".mid_block.attentions.0", h, g, eps)
    h = resnet_block(w, p + ".mid_block.resnets.1", h, None, g, eps)
    h = F.group_norm(h, g, w[p + ".conv_norm_out.weight"], w[p + ".conv_norm_out.bias"], eps)
    h = F.silu(h)
    return F.conv2d(h, w[p + ".conv_out.weight"], w[p + ".conv_out.bias"], padding=1)


def vae_encode_latent(w, cfg: dict, x: Tensor) -> Tensor:
    """meta_arch.py:142-145 / 209-212: encoder -> quant_conv -> mean half -> * scaling_factor."""
    h = vae_encoder(w, cfg, x)
    m = F.conv2d(h, w["vae.quant_conv.weight"], w["vae.quant_conv.bias"])
    mean, _ = torch.chunk(m, 2, dim=1)
    return mean * cfg["vae_scaling_factor"]


def vae_decoder(w, cfg: dict, z: Tensor) -> Tensor:
    """[3P AutoencoderKL.decoder] (Appendix A.6); call site meta_arch.py:256."""
    g, eps = cfg["vae_groups"], cfg["vae_eps"]
    rev = list(reversed(cfg["vae_channels"]))
    p = "vae.decoder"
    h = F.conv2d(z, w[p + ".conv_in.weight"], w[p + ".conv_in.bias"], padding=1)
    h = resnet_block(w, p + ".mid_block.resnets.0", h, None, g, eps)
    h = vae_attention(w, p + ".mid_block.attentions.0", h, g, eps)
    h = resnet_block(w, p + ".mid_block.resnets.1", h, None, g, eps)
    for i in range(len(rev)):
        for j in range(cfg["vae_layers_per_block"] + 1):
            h = resnet_block(w, f"{p}.up_blocks.{i}.resnets.{j}", h, None, g, eps)
        if i < len(rev) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, w[f"{p}.up_blocks.{i}.upsamplers.0.conv.weight"],
                         w[f"{p}.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    h = F.group_norm(h, g, w[p + ".conv_norm_out.weight"], w[p + ".conv_norm_out.bias"], eps)
    h = F.silu(h)
    return F.conv2d(h, w[p + ".conv_out.weight"], w[p + ".conv_out.bias"], padding=1)


# ----------------------------------------------------------------------------------------------
# U-Net (CustomUNet.forward, replace.py:379-549)
# ----------------------------------------------------------------------------------------------
def unet_embedding(w, cfg: dict, trans: Tensor, coords_emb: Tensor, cond_key: str = "bbox_mask_coords") -> Tensor:
    """replace.py:419-459: emb = time_embedding(time_proj(trans)) + bbox_embedding(coords) (or point_embedding for
    added_cond_kwargs["point_coords"], replace.py:446-450)."""
    B = trans.shape[0]
    op = get_timestep_embedding(trans, cfg["unet_channels"][0], True, 0.0)       # replace.py:432
    op = timestep_embedding_mlp(w, "unet.time_embedding", op)                     # replace.py:435
    ce = coords_emb.reshape(B, -1)                                                # replace.py:453
    aug = timestep_embedding_mlp(w, "unet.point_embedding" if cond_key == "point_coords" else "unet.bbox_embedding", ce)  # :450,455
    return op + aug                                                               # replace.py:459


def unet_forward(w, cfg: dict, sample: Tensor, trans: Tensor, ehs: Tensor, coords_emb: Tensor,
                 attention_mask: Optional[Tensor], taps: Optional[dict] = None, cond_key: str = "bbox_mask_coords") -> Tensor:
    uc = cfg["unet_channels"]
    heads = cfg["unet_heads"]
    g = cfg["unet_groups"]
    reps, geps, leps = cfg["unet_res_eps"], cfg["unet_tf_gn_eps"], cfg["unet_ln_eps"]
    nlev = len(uc)
    bias = None
    if attention_mask is not None:                                                # replace.py:401-403
        bias = (1 - attention_mask.to(sample.dtype)) * cfg["attn_mask_value"]
        bias = bias.unsqueeze(1)
    emb = unet_embedding(w, cfg, trans, coords_emb, cond_key)
    hw0 = tuple(sample.shape[-2:])                                               # level-0 token grid (square in the reference)
    h = F.conv2d(sample, w["unet.conv_in.weight"], w["unet.conv_in.bias"], padding=1)   # :462
    if taps is not None:
        taps["unet.conv_in"] = h
    skips = [h]
    for i in range(nlev):                                                         # replace.py:472-488
        for j in range(cfg["unet_layers_per_block"]):
            h = resnet_block(w, f"unet.down_blocks.{i}.resnets.{j}", h, emb, g, reps)
            if i < nlev - 1:
                h = transformer_2d(w, f"unet.down_blocks.{i}.attentions.{j}", h, ehs, bias, heads[i], g, geps, leps, hw0)
            skips.append(h)
        if i < nlev - 1:
            h = F.conv2d(h, w[f"unet.down_blocks.{i}.downsamplers.0.conv.weight"],
                         w[f"unet.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2, padding=1)
            skips.append(h)
        if taps is not None:
            taps[f"unet.down{i}"] = h
    h = resnet_block(w, "unet.mid_block.resnets.0", h, emb, g, reps)             # replace.py:493-504
    h = transformer_2d(w, "unet.mid_block.attentions.0", h, ehs, bias, heads[-1], g, geps, leps, hw0)
    h = resnet_block(w, "unet.mid_block.resnets.1", h, emb, g, reps)
    if taps is not None:
        taps["unet.mid"] = h
    rheads = list(reversed(heads))
    for i in range(nlev):                                                         # replace.py:509-536
        for j in range(cfg["unet_layers_per_block"] + 1):
            s = skips.pop()
            h = torch.cat([h, s], dim=1)
            h = resnet_block(w, f"unet.up_blocks.{i}.resnets.{j}", h, emb, g, reps)
            if i > 0:
                h = transformer_2d(w, f"unet.up_blocks.{i}.attentions.{j}", h, ehs, bias, rheads[i], g, geps, leps, hw0)
        if i < nlev - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, w[f"unet.up_blocks.{i}.upsamplers.0.conv.weight"],
                         w[f"unet.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
        if taps is not None:
            taps[f"unet.up{i}"] = h
    h = F.group_norm(h, g, w["unet.conv_norm_out.weight"], w["unet.conv_norm_out.bias"], reps)   # :541-544
    h = F.silu(h)
    return F.conv2d(h, w["unet.conv_out.weight"], w["unet.conv_out.bias"], padding=1)


# ----------------------------------------------------------------------------------------------
# SDMatte.forward (meta_arch.py:127-261).  Defaults = the trimap path configured by sdmatte_nodes.py:286-296;
# the keyword arguments restate the constructor options that select the other prompt types (meta_arch.py:31-77).
# ----------------------------------------------------------------------------------------------
AUX_INPUT_DIT = {"auto_mask": "auto_coords", "point_mask": "point_coords", "bbox_mask": "bbox_coords", "mask": "mask_coords",
                 "trimap": "trimap_coords"}                                       # meta_arch.py:22-28


def point_coords_embedding(coor: Tensor, P: int, use_coor_input: bool = True) -> Tensor:
    """meta_arch.py:152-176: pad the N point coordinates with zeros to the first i in [N, P) that divides P (P = 1680 in the
    reference, = point_embeddings_input_dim), embed every padded value with P // i channels."""
    B, N = coor.shape
    for i in range(N, P):
        if P % i == 0:
            coor = torch.cat([coor, torch.zeros(B, i - N, dtype=coor.dtype)], dim=1)
            if not use_coor_input:
                coor = torch.zeros_like(coor)                                     # :160,170-176
            return get_timestep_embedding(coor.flatten(), P // i, True, 0.0)
    raise ValueError(f"point prompt: {N} coordinates cannot be padded to a divisor of {P}")


@torch.no_grad()
def sdmatte_forward(w: Dict[str, Tensor], cfg: dict, data: dict, taps: Optional[dict] = None, aux_input: str = "trimap",
                    use_coor_input: bool = True, use_attention_mask: bool = True,
                    attn_mask_aux_input=("point_mask", "bbox_mask", "mask", "trimap")) -> Tensor:
    rgb = data["image"].float()                                                   # :128
    B = rgb.shape[0]
    aux = data[aux_input].float().repeat(1, 3, 1, 1)                              # :140-141
    aux_latent = vae_encode_latent(w, cfg, aux)                                   # :142-145
    coor_name = AUX_INPUT_DIT[aux_input]                                          # :150
    cond_key = "bbox_mask_coords"
    if coor_name == "point_coords":                                               # :152-176
        coor = point_coords_embedding(data[coor_name].float(), cfg["point_embeddings_input_dim"], use_coor_input)
        cond_key = "point_coords"
    else:                                                                         # :177-197
        coor = data[coor_name].float() if use_coor_input else torch.tensor([[0.0, 0.0, 1.0, 1.0]] * B)
        coor = get_timestep_embedding(coor.flatten(), cfg["bbox_embeddings_input_dim"] // 4, True, 0.0)  # :181-186
    attention_mask = None
    if use_attention_mask and aux_input in attn_mask_aux_input:                   # :200
        m = (data[aux_input].float() + 1) / 2                                     # :201-202
        m = F.interpolate(m, scale_factor=1 / 8, mode="nearest")                  # :203
        attention_mask = m.flatten(start_dim=1)                                   # :204
    rgb_latent = vae_encode_latent(w, cfg, rgb)                                   # :209-212
    ehs = F.conv2d(aux_latent, w["unet.aux_conv_in.weight"], w["unet.aux_conv_in.bias"], padding=1)  # :216
    ehs = ehs.view(B, cfg["cross_attention_dim"], -1).permute(0, 2, 1)            # :217-218
    trans = 1 - data["is_trans"]                                                  # :237-238
    unet_in = torch.cat([rgb_latent, aux_latent], dim=1)                          # :244
    if taps is not None:
        taps["aux_latent"], taps["rgb_latent"], taps["ehs"] = aux_latent, rgb_latent, ehs
        taps["attention_mask"] = attention_mask
    lat = unet_forward(w, cfg, unet_in, trans, ehs, coor, attention_mask, taps, cond_key)   # :245-253
    if taps is not None:
        taps["unet_out"] = lat
    lat = lat / cfg["vae_scaling_factor"]                                         # :254
    z = F.conv2d(lat, w["vae.post_quant_conv.weight"], w["vae.post_quant_conv.bias"])  # :255
    stacked = vae_decoder(w, cfg, z)                                              # :256
    if taps is not None:
        taps["decoded"] = stacked
    mean = stacked.mean(dim=1, keepdim=True)                                      # :258
    out = torch.clip(mean, -1.0, 1.0)                                             # :259
    return (out + 1.0) / 2.0                                                      # :260


# ----------------------------------------------------------------------------------------------
# Node pre/post-processing (sdmatte_nodes.py:204-214, 339-397)
# ----------------------------------------------------------------------------------------------
def resize_bilinear_aa(x: Tensor, size_hw) -> Tensor:
    """[3P torchvision.transforms.Resize on a tensor] = F.interpolate(bilinear, align_corners=False,
    antialias=True) (identity when sizes match)."""
    if tuple(x.shape[-2:]) == tuple(size_hw):
        return x
    return F.interpolate(x, size=tuple(size_hw), mode="bilinear", align_corners=False, antialias=True)


def preprocess(image_bhwc: Tensor, trimap_bhw: Tensor, inference_size: int, is_transparent: bool) -> dict:
    """sdmatte_nodes.py:339-353."""
    B = image_bhwc.shape[0]
    S = int(inference_size)
    img = image_bhwc.permute(0, 3, 1, 2).contiguous()
    img = resize_bilinear_aa(img, (S, S))
    img = (img - 0.5) / 0.5
    tri = resize_bilinear_aa(trimap_bhw.unsqueeze(1).contiguous(), (S, S)) * 2 - 1
    return {
        "image": img,
        "is_trans": torch.tensor([1 if is_transparent else 0] * B),
        "caption": [""] * B,
        "trimap": tri,
        "trimap_coords": torch.tensor([[0, 0, 1, 1]] * B, dtype=tri.dtype),
    }


def postprocess(pred_alpha_b1ss: Tensor, image_bhwc: Tensor, trimap_bhw: Tensor, output_mode: str,
                mask_refine: bool, trimap_constraint: float):
    """sdmatte_nodes.py:362-405."""
    H, W = image_bhwc.shape[1:3]
    out = resize_bilinear_aa(pred_alpha_b1ss, (H, W))
    out = out.squeeze(1).clamp(0, 1)
    if mask_refine:                                                               # :365-380
        fg = trimap_bhw > trimap_constraint
        bg = trimap_bhw < (1.0 - trimap_constraint)
        unk = ~(fg | bg)
        a = out.clone()
        a[bg] = 0.0
        a[fg] = torch.clamp(a[fg] * 1.2, 0, 1)
        a[(a < 0.3) & unk] = 0.0
        out = a
    ae = out.unsqueeze(-1)
    if output_mode == "alpha_only":                                               # :384-397
        matted = torch.zeros_like(image_bhwc)
    elif output_mode == "matted_rgba":
        matted = torch.cat([image_bhwc, ae], dim=-1)
    elif output_mode == "matted_rgb":
        fgm = (trimap_bhw.unsqueeze(-1) > 0.2) & (ae > 0.1)
        matted = image_bhwc * fgm.float()
    else:
        matted = image_bhwc * ae
    return out, matted


@torch.no_grad()
def apply_matte(w, cfg: dict, image_bhwc: Tensor, trimap_bhw: Tensor, inference_size: int,
                is_transparent: bool = False, output_mode: str = "alpha_only", mask_refine: bool = True,
                trimap_constraint: float = 0.8):
    """End-to-end oracle of SDMatteApply.apply_matte with force_cpu=True semantics."""
    data = preprocess(image_bhwc.float(), trimap_bhw.float(), inference_size, is_transparent)
    pred = sdmatte_forward(w, cfg, data)
    return postprocess(pred, image_bhwc.float(), trimap_bhw.float(), output_mode, mask_refine, trimap_constraint)


# ----------------------------------------------------------------------------------------------
# weight surgery helpers (utils.py:13-41) - restated for the G4 fixture
# ----------------------------------------------------------------------------------------------
def conv_in_surgery(weight_o4hw: Tensor, bias_o: Tensor, num: int):
    """replace_unet_conv_in (utils.py:13-30): tile the 4-ch conv_in `num` times, divide by num."""
    return weight_o4hw.repeat((1, num, 1, 1)) / num, bias_o.clone()


def aux_conv_in_init(weight_o4hw: Tensor, bias_o: Tensor, out_channels: int = 1024):
    """add_aux_conv_in (utils.py:33-41): first 320 filters copy conv_in, the rest are zero."""
    o = weight_o4hw.shape[0]
    wt = torch.zeros(out_channels, 4, 3, 3)
    bs = torch.zeros(out_channels)
    wt[:o] = weight_o4hw
    bs[:o] = bias_o
    return wt, bs
