"""Run the real kernel sources on the CPU fiber emulator (tests/emu) at tiny shapes.  This validates
index arithmetic, LDS layouts, barrier placement and MFMA fragment usage in the GPU-less build container;
the authoritative parity tests are the `-m gpu` ones (tests/test_gpu_*.py) which run the hipcc build."""
import ctypes
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ops_suite as S  # noqa: E402


@pytest.fixture(scope="module")
def emu_engine(pkg):
    from emu.build_emu import build
    from comfyui_sdmatte_amd.engine import Bindings, Engine
    from comfyui_sdmatte_amd.config import SDMatteConfig
    lib = Bindings(ctypes.CDLL(build()))
    eng = Engine(SDMatteConfig.tiny(), 0, True, _lib=lib)
    yield eng
    eng.close()


DEV = "cpu"


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5])
def test_conv3x3_s1_all_tile_cfgs(emu_engine, cfg):
    cin = 32 if cfg == 1 else 16
    S.check_conv(emu_engine, DEV, 1, 9, 35, cin, 40, tile_cfg=cfg, seed=cfg)


def test_conv3x3_split_precision_and_fused_groupnorm(emu_engine):
    """Precise-mode kernels (split-fp16 operands, half-plane LDS layout) and the fused GroupNorm+SiLU staging at op level."""
    for cfg in (0, 5, 2):
        S.check_conv(emu_engine, DEV, 1, 9, 35, 32, 40, tile_cfg=cfg, in_f32=True, out_f32=True, split=True, res="f32", seed=40 + cfg, atol=2e-5)
    for cfg in (0, 4, 5):
        S.check_conv(emu_engine, DEV, 1, 9, 35, 32, 40 if cfg != 4 else 3, tile_cfg=cfg, in_f32=True, out_f32=True, split=True, gn=(1e-6, True),
                     seed=50 + cfg, atol=2e-5)
        S.check_conv(emu_engine, DEV, 1, 9, 35, 32, 40 if cfg != 4 else 3, tile_cfg=cfg, in_f32=(cfg != 0), out_f32=True, gn=(1e-5, cfg != 5),
                     seed=60 + cfg)
    S.check_conv(emu_engine, DEV, 1, 7, 11, 64, 72, ntaps=1, tile_cfg=4, in_f32=True, out_f32=True, split=True, seed=70, atol=2e-5)      # KC=32 fp32 GEMM tile
    S.check_conv(emu_engine, DEV, 1, 5, 13, 64, 128, ntaps=1, geglu=True, tile_cfg=1, in_f32=True, out_f32=True, split=True, seed=71, atol=5e-5)
    S.check_conv(emu_engine, DEV, 2, 5, 13, 64, 96, ntaps=1, tile_cfg=2, in_f32=True, out_f32=True, res="f32", seed=72)                 # batch of 2 (image-aligned row tiles are an engine path; plain here)


def test_conv3x3_dma_weight_pipeline(emu_engine):
    """The DMA-weight kernels (256x128 tile, Cout >= 128: stage-ordered weights through the 4-stage LDS ring, double-buffered /
    split activation tiles): 1 and several K-chunks, a ragged second output-channel tile, fp16 / fp32 / fused GroupNorm / split
    precision, nearest-upsample + concat."""
    S.check_conv(emu_engine, DEV, 1, 9, 35, 16, 128, tile_cfg=0, seed=1)
    S.check_conv(emu_engine, DEV, 2, 10, 40, 48, 160, tile_cfg=0, res="f32", out_f32=True, seed=2)
    S.check_conv(emu_engine, DEV, 1, 9, 35, 32, 128, tile_cfg=0, in_f32=True, out_f32=True, seed=3)
    S.check_conv(emu_engine, DEV, 1, 9, 35, 64, 128, tile_cfg=0, in_f32=True, out_f32=True, gn=(1e-6, True), seed=4)
    S.check_conv(emu_engine, DEV, 1, 17, 33, 32, 128, tile_cfg=0, in_f32=False, out_f32=True, gn=(1e-6, True), seed=5)
    S.check_conv(emu_engine, DEV, 1, 9, 35, 16, 128, tile_cfg=0, in_f32=True, out_f32=True, split=True, seed=6, atol=2e-5)
    S.check_conv(emu_engine, DEV, 1, 9, 35, 96, 160, tile_cfg=0, in_f32=True, out_f32=True, split=True, gn=(1e-6, True), seed=7, atol=2e-5)
    S.check_conv(emu_engine, DEV, 1, 6, 20, 32, 128, C1=32, up=1, tile_cfg=0, in_f32=True, out_f32=True, split=True, seed=8, atol=2e-5)


def test_conv3x3_producer_consumer_form(emu_engine, engine_option):
    """PC form of the split-precision DMA-weight kernel (8 waves: 4 MFMA-only consumer waves + 4 staging producer waves, A tile
    double-buffered, weights two stages ahead in a ring of 5): 1 / odd / even numbers of K-chunks, ragged output-channel tile, fused
    GroupNorm, upsample + concat, residual; and agreement with the 4-wave form up to the fp32 summation order."""
    engine_option(emu_engine, "conv_pc", 1)
    S.check_conv(emu_engine, DEV, 1, 9, 35, 16, 128, tile_cfg=0, in_f32=True, out_f32=True, split=True, seed=6, atol=2e-5)
    S.check_conv(emu_engine, DEV, 2, 9, 35, 96, 160, tile_cfg=0, in_f32=True, out_f32=True, split=True, gn=(1e-6, True), res="f32", seed=7, atol=2e-5)
    S.check_conv(emu_engine, DEV, 1, 6, 20, 32, 128, C1=32, up=1, tile_cfg=0, in_f32=True, out_f32=True, split=True, seed=8, atol=2e-5)
    torch.manual_seed(3)
    x = torch.randn(1, 10, 33, 80)
    w = torch.randn(128, 80, 3, 3) / 27.0
    a = emu_engine.op_conv(x, w, None, out_f32=True, split=True, tile_cfg=0)
    engine_option(emu_engine, "conv_pc", 0)
    b = emu_engine.op_conv(x, w, None, out_f32=True, split=True, tile_cfg=0)
    assert (a - b).abs().max().item() <= 2e-5


def test_conv3x3_fp8_residual_terms(emu_engine):
    """F8 kernel: x_hi.w_hi on fp16, the residual terms x_lo.w and x.w_lo on e4m3 operands (one K=64 MFMA per tap and 32 channels),
    32-channel chunks, producer / consumer waves.  1 / 2 / 3 chunks, ragged output-channel tile, fused GroupNorm + SiLU, residual,
    upsample + concat.  Error bound: a residual term is 2^-11 of the product and e4m3 rounds at 2^-4: ~2^-15 per operand."""
    e0 = S.check_conv(emu_engine, DEV, 1, 9, 35, 32, 128, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, seed=6, atol=3e-4)
    S.check_conv(emu_engine, DEV, 2, 9, 35, 96, 160, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, gn=(1e-6, True), res="f32", seed=7, atol=3e-4)
    S.check_conv(emu_engine, DEV, 1, 6, 20, 32, 128, C1=32, up=1, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, seed=8, atol=3e-4)
    assert e0 < 1e-4, e0          # plain fp16 operands sit at ~1.5e-3 against the un-rounded reference on these inputs
    # a concat boundary inside a 32-channel chunk (C0 = 48): the layer falls back to the register-staged split kernel
    S.check_conv(emu_engine, DEV, 1, 6, 20, 48, 128, C1=16, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, seed=9, atol=3e-5)


@pytest.mark.parametrize("mode", ["0", "3", "4"])
def test_f8_accumulator_layout_epilogue(emu_engine, engine_option, mode):
    """F8 kernels ([channel][pixel] accumulators), full fp32 tiles: 16-byte stores straight from the MFMA registers + bias from LDS
    + the residual as the accumulators' initial value (option conv_epi = 4, the default), the residual init alone (=3) and the LDS-staged
    epilogue (=0): same results on full tiles (3x3: H % 8 == 0 and W % 32 == 0; GEMM: rows % 256 == 0), a ragged output-channel
    tile, with and without residual, fused GroupNorm; a ragged image mixes both paths in one launch."""
    engine_option(emu_engine, "conv_epi", int(mode))
    S.check_conv(emu_engine, DEV, 1, 16, 64, 64, 160, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, res="f32", seed=71, atol=3e-4)
    S.check_conv(emu_engine, DEV, 2, 8, 32, 32, 128, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, gn=(1e-6, True), seed=72, atol=3e-4)
    S.check_conv(emu_engine, DEV, 1, 12, 40, 32, 128, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, res="f32", seed=73, atol=3e-4)
    S.check_conv(emu_engine, DEV, 1, 16, 32, 64, 200, ntaps=1, tile_cfg=4, in_f32=True, out_f32=True, split=True, f8=True, res="f32", seed=74, atol=3e-4)
    S.check_conv(emu_engine, DEV, 1, 8, 32, 32, 128, ntaps=1, tile_cfg=4, in_f32=True, out_f32=True, split=True, f8=True, seed=75, atol=3e-4)


def test_f8_residual_operand_ranges(emu_engine):
    """The fp8 residual operands must not saturate on real-checkpoint ranges: activations x300 (e5m2 operands: the range of fp16),
    weights x4 and x1/64 (e4m3 with the layer's own power-of-two scale, chosen from max|w| when the layer is packed) stay at the
    ~2^-14 relative level of the arithmetic; with fixed scales the residual terms clamp and the result drops to the fp16-operand
    level (~1e-3 relative)."""
    for xs, ws, seed in ((300.0, 1.0, 91), (1.0, 4.0, 92), (300.0, 4.0, 93), (0.01, 1.0 / 64, 94)):
        e = S.check_conv(emu_engine, DEV, 1, 8, 32, 64, 128, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, seed=seed, atol=3e-4,
                         xscale=xs, wscale=ws, rel=True)
        assert e < 1e-4, (xs, ws, e)
    S.check_conv(emu_engine, DEV, 1, 8, 32, 64, 128, ntaps=1, tile_cfg=4, in_f32=True, out_f32=True, split=True, f8=True, res="f32", seed=95, atol=3e-4,
                 xscale=300.0, wscale=4.0, rel=True)


@pytest.mark.parametrize("tile", [256, 128, 64])
def test_gemm_p3_plane_fed_gemm(emu_engine, engine_option, tile):
    """k_gemm.h on the emulator: every epilogue (fp32 + residual, GEGLU -> planes, linear -> planes, attention operand planes, fused statistics),
    1 / 2 / 3 K chunks, ragged row tiles (rows % 64 != 0), ragged / several output-channel tiles, a batch whose images do not fill a row tile,
    LayerNorm with plane output in front.  Error bound: the residual terms are 2^-11 of the product on e5m2 x e4m3 operands (and e5m2(x) by truncation)."""
    so = engine_option
    e0 = S.check_gemm_p3(emu_engine, DEV, 1, 5, 13, 64, 96, mode=0, tile=tile, seed=1, set_option=so)
    assert e0 < 1e-4, e0
    S.check_gemm_p3(emu_engine, DEV, 2, 6, 11, 32, 160, mode=0, res=True, tile=tile, seed=2, set_option=so)
    S.check_gemm_p3(emu_engine, DEV, 1, 9, 30, 96, 256, mode=1, tile=tile, seed=3, set_option=so)                      # GEGLU: 128 outputs, 3 chunks, 270 rows
    S.check_gemm_p3(emu_engine, DEV, 1, 8, 16, 64, 64, mode=3, res=True, tile=tile, seed=4, set_option=so)
    S.check_gemm_p3(emu_engine, DEV, 1, 7, 9, 64, 192, mode=2, lo_cols=128, tile=tile, seed=5, set_option=so)          # q | k | v: no pair plane for the V third
    S.check_gemm_p3(emu_engine, DEV, 2, 8, 12, 64, 96, mode=4, res=True, tile=tile, seed=6, set_option=so)             # 96 rows per image: image-aligned ragged tiles
    S.check_gemm_p3(emu_engine, DEV, 1, 4, 40, 64, 96, mode=0, ln=True, tile=tile, seed=7, set_option=so)


@pytest.mark.parametrize("grid", [3, 8])
def test_gemm_p3_persistent_blocks_walk_several_tiles(emu_engine, engine_option, grid):
    """a forced small grid: every block multiplies several tiles as one DMA stream (the next tile's first chunk is issued during the current tile's last
    chunk), with the XCD-aware tile order (>= 8 row tiles) and without, one and several chunks per tile, and every epilogue that keeps per-tile state"""
    so = engine_option
    so(emu_engine, "gemm_p3_persist", grid)
    S.check_gemm_p3(emu_engine, DEV, 1, 9, 30, 32, 256, mode=0, res=True, tile=64, seed=21, set_option=so)          # 5 x 2 tiles, 1 chunk each
    S.check_gemm_p3(emu_engine, DEV, 1, 16, 33, 96, 192, mode=3, res=True, tile=64, seed=22, set_option=so)         # 9 row tiles: XCD order, padding ids
    S.check_gemm_p3(emu_engine, DEV, 3, 8, 12, 64, 160, mode=4, tile=64, seed=23, set_option=so)                    # image-aligned tiles + statistics
    S.check_gemm_p3(emu_engine, DEV, 1, 10, 32, 64, 512, mode=1, tile=128, seed=24, set_option=so)


def test_gemm_p3_operand_ranges(emu_engine):
    """activations x300 / weights x4 and x1/64: the e5m2 operands (range of fp16) and the per-layer e4m3 weight scale keep the relative error at the level of the arithmetic"""
    for xs, ws, seed in ((300.0, 4.0, 11), (0.01, 1.0 / 64, 12)):
        e = S.check_gemm_p3(emu_engine, DEV, 1, 4, 32, 64, 128, mode=0, seed=seed, xscale=xs, wscale=ws, rel=True)
        assert e < 1e-4, (xs, ws, e)


def test_conv3x3_thin_output_tile(emu_engine):
    S.check_conv(emu_engine, DEV, 2, 10, 33, 32, 3, in_f32=True, tile_cfg=4, seed=9)          # conv_out shape: Cout 3 -> one 32-wide tile


def test_conv3x3_s1_multi_chunk_and_batch(emu_engine):
    S.check_conv(emu_engine, DEV, 2, 12, 12, 64, 96, tile_cfg=2, res="f32", out_f32=True, seed=5)


@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
@pytest.mark.parametrize("pad_mode", [0, 1])
def test_conv3x3_s2(emu_engine, cfg, pad_mode):
    S.check_conv(emu_engine, DEV, 1, 16, 40, 16, 32, stride=2, pad_mode=pad_mode, tile_cfg=cfg, seed=7 + cfg)


def test_conv3x3_upsample_concat_fp32in(emu_engine):
    S.check_conv(emu_engine, DEV, 1, 6, 10, 32, 32, up=1, tile_cfg=2, seed=11)
    S.check_conv(emu_engine, DEV, 1, 8, 9, 32, 64, C1=32, in_f32=True, tile_cfg=2, res="f16", seed=12)


@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
def test_gemm_all_tile_cfgs(emu_engine, cfg):
    cin = 16 if cfg == 3 else 64
    S.check_conv(emu_engine, DEV, 1, 7, 11, cin, 72, ntaps=1, tile_cfg=cfg, res="f32", out_f32=True, seed=20 + cfg)


def test_gemm_fp8_residual_terms(emu_engine):
    """Linear / 1x1 GEMM on the 8-wave fp8-residual kernel (F8 with NTAPS = 1, tile cfg 4): 1 / 2 / 5 chunks of 32 channels, ragged
    rows and output-channel tiles, residual, GEGLU epilogue, several tiles per block."""
    S.check_conv(emu_engine, DEV, 1, 7, 11, 32, 128, ntaps=1, tile_cfg=4, in_f32=True, out_f32=True, split=True, f8=True, seed=80, atol=3e-4)
    S.check_conv(emu_engine, DEV, 2, 19, 23, 64, 200, ntaps=1, tile_cfg=4, in_f32=True, out_f32=True, split=True, f8=True, res="f32", seed=81, atol=3e-4)
    S.check_conv(emu_engine, DEV, 1, 9, 33, 160, 256, ntaps=1, geglu=True, tile_cfg=4, in_f32=True, out_f32=True, split=True, f8=True, seed=82, atol=3e-4)


def test_gemm_geglu_and_scale(emu_engine):
    S.check_conv(emu_engine, DEV, 1, 5, 13, 64, 256, ntaps=1, geglu=True, tile_cfg=2, seed=30)
    S.check_conv(emu_engine, DEV, 1, 5, 13, 64, 128, ntaps=1, geglu=True, tile_cfg=0, seed=31)
    S.check_conv(emu_engine, DEV, 1, 4, 4, 16, 8, ntaps=1, out_scale=0.18215, tile_cfg=3, seed=32)


def test_groupnorm(emu_engine):
    S.check_groupnorm(emu_engine, DEV, 2, 9, 7, 64, in_f32=True, silu=True)
    S.check_groupnorm(emu_engine, DEV, 1, 5, 5, 320, in_f32=False, silu=False, eps=1e-5)
    S.check_groupnorm(emu_engine, DEV, 1, 6, 6, 128, C1=64, in_f32=True, silu=True)


def test_layernorm(emu_engine):
    S.check_layernorm(emu_engine, DEV, 37, 128, in_f32=True)
    S.check_layernorm(emu_engine, DEV, 9, 320, in_f32=False)


def test_attention_d64(emu_engine):
    S.check_attention(emu_engine, DEV, 1, 2, 70, 100, 64, use_bias=True, fused_stride=True)
    S.check_attention(emu_engine, DEV, 2, 1, 33, 64, 64, use_bias=False)
    S.check_attention(emu_engine, DEV, 1, 1, 32, 192, 64, use_bias=False, spike=True, seed=3)


def test_attention_d64_split_precision(emu_engine, engine_option):
    """The default precision's attention cores: Q.K^T on split operands (the logits feed an exponential), P.V on plain fp16 operands
    (PREC = 2), fp32 in / out; and the fully split form (PREC = 1, option attn_pv_split).  Against un-rounded fp64 attention: the
    fp16-operand kernel sits at ~2e-3 on these inputs."""
    e2 = S.check_attention(emu_engine, DEV, 1, 2, 70, 100, 64, use_bias=True, split=True, atol=1e-3)
    S.check_attention(emu_engine, DEV, 2, 1, 33, 200, 64, use_bias=False, split=True, seed=4, atol=1e-3)
    engine_option(emu_engine, "attn_pv_split", 1)
    e1 = S.check_attention(emu_engine, DEV, 1, 2, 70, 100, 64, use_bias=True, split=True, atol=3e-5)
    assert e1 < e2
    engine_option(emu_engine, "attn_pv_split", 0)
    # the residual terms of Q.K^T on fp8 MFMAs (PREC = 3, the default) against the same kernel with fp16 residual terms (option attn_f8 = 0,
    # PREC = 2): P.V is rounded identically in both, so the difference isolates the logit error of the e5m2 residual pairs
    import torch
    g = torch.Generator().manual_seed(21)
    q, k, v = torch.randn(1, 70, 128, generator=g) * 1.5, torch.randn(1, 130, 128, generator=g) * 1.5, torch.randn(1, 130, 128, generator=g)
    a8 = emu_engine.op_attention_split(q, k, v, 2)
    engine_option(emu_engine, "attn_f8", 0)
    a16 = emu_engine.op_attention_split(q, k, v, 2)
    d = (a8 - a16).abs().max().item()
    assert 0.0 < d < 4e-4, d            # logit spread ~3x that of unit-variance q / k; fp16 operands alone are off by ~1e-2 here
    engine_option(emu_engine, "attn_f8", 1)
    # the 8-wave form of the same kernel (level-0 attentions; global loads two key tiles ahead through two raw-tile register sets): same
    # arithmetic per query row as the 4-wave form -> bit-identical, for 1, 2, 3 and 5 key tiles and the trimap-style tile list
    for lk in (40, 64, 128, 130, 320, 450, 448):
        kk, vv = torch.randn(1, lk, 128, generator=g) * 1.5, torch.randn(1, lk, 128, generator=g)
        bias = torch.where(torch.rand(1, lk, generator=g) < 0.3, torch.tensor(-10000.0), torch.tensor(0.0))
        for bb in (None, bias):
            engine_option(emu_engine, "attn_nw", 4)
            r4 = emu_engine.op_attention_split(q, kk, vv, 2, bias=bb)
            engine_option(emu_engine, "attn_pipe4", 1)          # the 4-wave pipeline (two K / three V^T buffers; off by default, unmeasured)
            r4p = emu_engine.op_attention_split(q, kk, vv, 2, bias=bb)
            engine_option(emu_engine, "attn_pipe4", 0)
            assert torch.equal(r4p, r4), (lk, bb is not None, (r4p - r4).abs().max().item())
            engine_option(emu_engine, "attn_nw", 8)
            engine_option(emu_engine, "attn_pipe", 0)
            r8 = emu_engine.op_attention_split(q, kk, vv, 2, bias=bb)
            assert torch.equal(r4, r8), (lk, bb is not None, (r4 - r8).abs().max().item())
            # the two-tile software pipeline of the 8-wave kernel (the default for 8-wave launches; three LDS buffers): same arithmetic
            engine_option(emu_engine, "attn_pipe", 1)
            rp = emu_engine.op_attention_split(q, kk, vv, 2, bias=bb)
            assert torch.equal(rp, r8), (lk, bb is not None, (rp - r8).abs().max().item())
            # the ping-pong of the block's two wave halves (attn_d64_pp_kernel; Lk % 64 == 0 only: tiles by LDS-DMA into unpadded, swizzled slots, K rows
            # permuted so that a V^T fragment is one 16-byte read - the k-slots of the P.V MFMAs are permuted with them): the same products, summed in another
            # order inside the MFMAs -> equal to the pipelines to fp32 rounding, not bit for bit
            engine_option(emu_engine, "attn_nw", 0)
            engine_option(emu_engine, "attn_pp", 1)
            engine_option(emu_engine, "attn_pp_min_blocks", 0)      # (the engine keeps launches of < 128 blocks on the pipelines)
            emu_engine.lib.kernel_counts(reset=True)
            rpp = emu_engine.op_attention_split(q, kk, vv, 2, bias=bb)
            assert emu_engine.lib.kernel_counts().get("attn_d64_pp", 0) == (1 if lk % 64 == 0 else 0)
            assert (rpp - r8).abs().max().item() <= 2e-6, (lk, bb is not None, (rpp - r8).abs().max().item())
            engine_option(emu_engine, "attn_pp", 0)
    engine_option(emu_engine, "attn_nw", 0)


def test_attention_d64_skips_underflowing_key_tiles(emu_engine, engine_option):
    # trimap-like bias with whole key tiles at -5000 / -10000: those tiles are never loaded; the result must equal the fp32
    # reference (where their probabilities underflow to 0) AND be bit-identical to walking every tile
    S.check_attention(emu_engine, DEV, 3, 2, 40, 500, 64, use_bias=True, blocks=True, seed=7)
    import torch
    g = torch.Generator().manual_seed(11)
    q, k, v = (torch.randn(2, n, 64, generator=g).half() for n in (50, 320, 320))
    bias = torch.full((2, 320), -10000.0)
    bias[0, 130:150] = 0.0
    bias[1, 5:9] = 0.0
    bias[1, 200:] = -5000.0
    sparse = emu_engine.op_attention(q, k, v, 1, bias)
    engine_option(emu_engine, "attn_dense", 1)
    dense = emu_engine.op_attention(q, k, v, 1, bias)
    assert torch.equal(sparse, dense)
    # the same property of the ping-pong kernel (split-precision operands; tile-list walk through scalar loads, dense walk by arithmetic), with and without a key split
    engine_option(emu_engine, "attn_pp", 1)
    engine_option(emu_engine, "attn_pp_min_blocks", 0)
    for ns in (1, 2):
        engine_option(emu_engine, "attn_ksplit", ns)
        engine_option(emu_engine, "attn_dense", 0)
        emu_engine.lib.kernel_counts(reset=True)
        sp = emu_engine.op_attention_split(q.float(), k.float(), v.float(), 1, bias=bias)
        assert emu_engine.lib.kernel_counts().get("attn_d64_pp", 0) == 1
        engine_option(emu_engine, "attn_dense", 1)
        de = emu_engine.op_attention_split(q.float(), k.float(), v.float(), 1, bias=bias)
        assert torch.equal(sp, de), (ns, (sp - de).abs().max().item())
    ref = torch.softmax((q.double() @ k.double().transpose(1, 2)) / 8.0 + bias.double()[:, None, :], -1) @ v.double()
    assert (de.double() - ref).abs().max().item() < 1e-3


def test_attention_d64_key_split(emu_engine, engine_option):
    """AttnParams::nsplit (engine option attn_ksplit): block row y walks its range of key tiles and leaves unnormalised partial sums, attn_combine_kernel
    merges them.  Against fp64 attention, close to the unsplit kernel, and - the ranges being ranges of KEYS - still bit-identical between the walk
    along the active-tile list and the dense walk; a range without any active tile (an empty part) included."""
    import torch
    for ns in (2, 4):
        engine_option(emu_engine, "attn_ksplit", ns)
        emu_engine.lib.kernel_counts(reset=True)
        S.check_attention(emu_engine, DEV, 2, 2, 40, 500, 64, use_bias=True, blocks=True, seed=7, split=True, atol=1e-3)
        S.check_attention(emu_engine, DEV, 1, 1, 70, 200, 64, use_bias=False, seed=8, split=True, atol=1e-3)
        assert emu_engine.lib.kernel_counts().get("attn_combine", 0) == 2
    g = torch.Generator().manual_seed(12)
    q, k, v = (torch.randn(2, n, 64, generator=g) for n in (50, 640, 640))
    bias = torch.full((2, 640), -10000.0)
    bias[0, 130:150] = 0.0          # image 0: active keys in the first quarter only -> three empty parts at nsplit = 4
    bias[1, 5:9] = 0.0
    bias[1, 400:] = -5000.0
    outs = {}
    for ns in (1, 4):
        engine_option(emu_engine, "attn_ksplit", ns)
        engine_option(emu_engine, "attn_dense", 0)
        sparse = emu_engine.op_attention_split(q, k, v, 1, bias)
        engine_option(emu_engine, "attn_dense", 1)
        dense = emu_engine.op_attention_split(q, k, v, 1, bias)
        assert torch.equal(sparse, dense), ns
        outs[ns] = sparse
    assert (outs[1] - outs[4]).abs().max().item() < 2e-6


def test_attention_d512(emu_engine):
    S.check_attention(emu_engine, DEV, 1, 1, 40, 64, 512, use_bias=False, atol=5e-3)
    S.check_attention(emu_engine, DEV, 1, 1, 33, 50, 512, use_bias=False, atol=5e-3, seed=2)   # ragged last key tile (clamped DMA rows + mask)


def test_resize_aa(emu_engine):
    S.check_resize(emu_engine, DEV, 2, 37, 53, 64, 64)
    S.check_resize(emu_engine, DEV, 1, 64, 64, 37, 53)
    S.check_resize(emu_engine, DEV, 1, 100, 30, 16, 24)


def test_emulated_e4m3_conversion_matches_torch_and_the_hardware_probe():
    """The emulator's v_cvt_pk_fp8_f32 stand-in (round to nearest even, subnormals, saturation at 448, NaN beyond the rounding range)
    against torch's float8_e4m3fn on a dense sweep, and against the bytes the MI355X produced for the probe's inputs
    (tools/probe/f8_semantics_probe.hip -> profiles/r02_f8_semantics_probe.txt)."""
    from emu.build_emu import build
    lib = ctypes.CDLL(build())
    lib.sdm_emu_f32_to_e4m3.argtypes = [ctypes.c_float]
    lib.sdm_emu_f32_to_e4m3.restype = ctypes.c_int
    lib.sdm_emu_e4m3_to_f32.argtypes = [ctypes.c_int]
    lib.sdm_emu_e4m3_to_f32.restype = ctypes.c_float
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(4000, generator=g) * s for s in (1e-3, 0.05, 1.0, 30.0, 200.0)] + [torch.linspace(-448, 448, 3585)])
    x = x.clamp(-448, 448)
    want = x.to(torch.float8_e4m3fn).view(torch.uint8)
    got = torch.tensor([lib.sdm_emu_f32_to_e4m3(float(v)) for v in x], dtype=torch.uint8)
    same = (got == want) | ((x == 0) & ((got & 0x7f) == 0) & ((want & 0x7f) == 0))          # +-0
    assert bool(same.all()), (x[~same][:5], got[~same][:5], want[~same][:5])
    for v in range(256):                                                                    # decode round trip (0x7f / 0xff are NaN)
        if (v & 0x7f) != 0x7f:
            f = lib.sdm_emu_e4m3_to_f32(v)
            assert lib.sdm_emu_f32_to_e4m3(f) == v or f == 0.0
    hw = {1.0: 0x38, -1.0: 0xb8, 0.3: 0x2a, 448.0: 0x7e, 449.0: 0x7e, 1000.0: 0x7f, 0.0625: 0x18, 0.001: 0x01, 1.0625: 0x38, 1.1875: 0x3a, 17.0: 0x58,
          0.0019: 0x01, 464.0: 0x7e, 3.3e-3: 0x02}
    for f, b in hw.items():
        assert lib.sdm_emu_f32_to_e4m3(f) == b, (f, hex(lib.sdm_emu_f32_to_e4m3(f)), hex(b))


def test_fp8_residual_kernels_random_shapes(emu_engine):
    """Seeded random shapes through the 8-wave fp8-residual conv / GEMM kernels: ragged sizes, batches, concat on chunk boundaries,
    fused GroupNorm (+SiLU), upsample, residual, ragged output-channel tiles, GEGLU (a 36-case sweep of the same generator ran
    clean when the kernels were written; 8 cases here)."""
    import random
    rng = random.Random(7)
    for it in range(6):
        N = rng.choice([1, 2, 3]); H = rng.randint(3, 30); W = rng.randint(5, 45)
        cin = rng.choice([32, 64, 96, 160]); cout = rng.choice([128, 136, 200, 256, 320])
        up = rng.choice([0, 0, 0, 1]); c1 = rng.choice([0, 0, 32, 64])
        gn = rng.choice([None, (1e-6, True), (1e-5, False)]) if up == 0 else None
        res = rng.choice([None, "f32"])
        if up:
            H, W = max(3, H // 2), max(5, W // 2)
        S.check_conv(emu_engine, DEV, N, H, W, cin, cout, C1=c1, up=up, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, gn=gn, res=res,
                     seed=100 + it, atol=3e-4)
    for it in range(2):
        N = rng.choice([1, 2]); H = rng.randint(3, 40); W = rng.randint(5, 40)
        cin = rng.choice([32, 64, 160, 320]); cout = rng.choice([128, 200, 256, 384])
        S.check_conv(emu_engine, DEV, N, H, W, cin, cout, ntaps=1, tile_cfg=4, in_f32=True, out_f32=True, split=True, f8=True, res="f32", seed=200 + it, atol=3e-4)


def test_conv3x3_f8_tiles_back_to_back(emu_engine):
    """F8 3x3 kernel, several tiles per block (three on the emulator) with the cross-tile prefetch: the next tile's chunk 0 waits in the producers'
    registers across the tile boundary (its high planes are written during the last step of the previous tile, its fp8 image in front of the next
    tile's first barrier), even and odd chunk counts (odd: no prefetch, ring slots restart), fused GroupNorm, residual as accumulator init."""
    emu_engine.lib.kernel_counts(reset=True)
    S.check_conv(emu_engine, DEV, 2, 40, 96, 128, 128, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, gn=(1e-6, True), res="f32", seed=81, atol=3e-4)
    S.check_conv(emu_engine, DEV, 1, 24, 32, 192, 128, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, gn=(1e-5, False), seed=82, atol=3e-4)
    S.check_conv(emu_engine, DEV, 1, 24, 64, 96, 128, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, res="f32", seed=83, atol=3e-4)
    S.check_conv(emu_engine, DEV, 1, 16, 64, 256, 256, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, seed=84, atol=3e-4)
    assert emu_engine.lib.kernel_counts().get("conv3x3_f8<gn>", 0) == 2 and emu_engine.lib.kernel_counts().get("conv3x3_f8", 0) == 2


def test_conv_split_k(emu_engine, engine_option):
    """ConvParams::ksplit + splitk_reduce_kernel on the emulator (ops_suite.check_conv_splitk)."""
    S.check_conv_splitk(emu_engine, DEV, engine_option)


def test_conv_const_tiles_are_filled_not_multiplied(emu_engine):
    """Piecewise-constant input + class plane through the F8 3x3 kernel: bit-identical to multiplying every tile (ops_suite.check_conv_const_tiles)."""
    S.check_conv_const_tiles_are_really_left_out(emu_engine, DEV)
    S.check_conv_const_tiles(emu_engine, DEV)
    S.check_conv_const_tiles(emu_engine, DEV, N=1, H=40, W=96, Cin=64, Cout=128, gn=False, res=False, seed=8)
