"""-m gpu: every HIP kernel (through the C ABI of libsdmatte_hip.so) vs a torch fp32 reference of the same
op on the same seeded inputs, at realistic channel counts and for every compiled tile configuration."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ops_suite as S  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def eng(pkg):
    from comfyui_sdmatte_amd.engine import Engine
    from comfyui_sdmatte_amd.config import SDMatteConfig
    e = Engine(SDMatteConfig.tiny(), 0, True)
    yield e
    e.close()


@pytest.mark.parametrize("cfg,cin,cout,H,W", [(0, 128, 128, 40, 96), (0, 16, 128, 33, 70), (1, 320, 320, 16, 64), (2, 640, 200, 16, 16),
                                              (2, 16, 1024, 16, 16), (-1, 256, 256, 64, 64), (-1, 1280, 1280, 8, 8), (3, 128, 128, 40, 96), (3, 512, 200, 24, 40),
                                              (4, 128, 3, 70, 100), (4, 512, 8, 33, 40), (5, 320, 320, 32, 64), (5, 1280, 200, 16, 32)])
def test_conv3x3_s1(eng, cfg, cin, cout, H, W):
    S.check_conv(eng, DEV, 2, H, W, cin, cout, tile_cfg=cfg, seed=cfg + 5)


@pytest.mark.parametrize("cfg,pad_mode", [(0, 0), (0, 1), (1, 0), (1, 1), (2, 1), (3, 0), (3, 1), (-1, 1)])
def test_conv3x3_s2(eng, cfg, pad_mode):
    S.check_conv(eng, DEV, 2, 32, 64, 128, 128, stride=2, pad_mode=pad_mode, tile_cfg=cfg, seed=40 + cfg)


def test_conv3x3_fusions(eng):
    S.check_conv(eng, DEV, 1, 16, 24, 128, 128, up=1, seed=50)                                    # Upsample2D
    S.check_conv(eng, DEV, 2, 16, 16, 640, 320, C1=320, in_f32=False, seed=51)                    # skip concat
    S.check_conv(eng, DEV, 1, 32, 32, 128, 128, res="f32", out_f32=True, seed=52)                 # fp32 residual stream
    S.check_conv(eng, DEV, 1, 32, 32, 128, 128, res="f16", out_f32=False, seed=53)
    S.check_conv(eng, DEV, 1, 24, 40, 128, 3, out_f32=True, seed=54)                              # decoder conv_out
    S.check_conv(eng, DEV, 1, 16, 16, 512, 8, seed=55)                                            # encoder conv_out


@pytest.mark.parametrize("cfg,cin,cout,H,W,in_f32,silu", [(0, 128, 128, 40, 96, True, True), (0, 128, 128, 40, 96, False, True),
                                                            (0, 512, 256, 24, 40, True, True), (0, 320, 320, 33, 35, False, False),
                                                            (4, 128, 3, 70, 100, True, True), (4, 512, 8, 33, 40, False, True),
                                                            (5, 320, 320, 32, 64, True, True), (5, 640, 640, 16, 32, False, True)])
def test_conv3x3_fused_groupnorm(eng, cfg, cin, cout, H, W, in_f32, silu):
    """The two fused-GroupNorm instantiations of the 256-pixel tiles (48 % of the step time in round 1) at op level: GroupNorm(32)
    (+SiLU) applied inside the operand staging == F.group_norm -> F.silu -> fp16 -> conv."""
    S.check_conv(eng, DEV, 2, H, W, cin, cout, tile_cfg=cfg, in_f32=in_f32, gn=(1e-6, silu), res="f32" if cfg != 4 else None,
                 out_f32=True, seed=30 + cfg)


def test_conv3x3_fused_groupnorm_concat(eng):
    S.check_conv(eng, DEV, 1, 16, 32, 640, 320, C1=320, in_f32=True, gn=(1e-5, True), tile_cfg=0, out_f32=True, seed=37)    # up-block ResBlock conv1


@pytest.mark.parametrize("ntaps,cfg,cin,cout,H,W,gn", [(9, 0, 128, 128, 40, 96, True), (9, 0, 256, 128, 16, 64, False), (9, 1, 320, 320, 16, 64, False),
                                                        (9, 2, 640, 200, 16, 16, False), (9, 4, 128, 3, 40, 64, True), (9, 5, 320, 320, 32, 64, True), (1, -1, 320, 960, 1, 4096, False),
                                                        (1, 1, 1280, 1280, 1, 1024, False), (1, 2, 1024, 640, 1, 300, False), (1, 3, 16, 8, 1, 256, False)])
def test_conv_split_precision(eng, ntaps, cfg, cin, cout, H, W, gn):
    """Precise-mode kernels (split-fp16 operands, 3 MFMAs per product) against the fp32 reference on UN-rounded operands: the
    fp16-operand kernels sit at ~1e-3 on these shapes, the split kernels must be at the fp32-accumulation level."""
    S.check_conv(eng, DEV, 2 if ntaps == 9 else 1, H, W, cin, cout, ntaps=ntaps, tile_cfg=cfg, in_f32=True, out_f32=True, split=True,
                 gn=(1e-6, True) if gn else None, res="f32" if cout % 4 == 0 else None, seed=80 + cfg, atol=3e-5)


def test_conv_split_precision_s2_up_geglu(eng):
    S.check_conv(eng, DEV, 2, 32, 64, 128, 128, stride=2, pad_mode=1, in_f32=True, out_f32=True, split=True, seed=90, atol=3e-5)
    S.check_conv(eng, DEV, 1, 16, 24, 128, 128, up=1, in_f32=True, out_f32=True, split=True, seed=91, atol=3e-5)
    S.check_conv(eng, DEV, 1, 1, 1024, 320, 2560, ntaps=1, geglu=True, in_f32=True, out_f32=True, split=True, seed=92, atol=1e-4)


@pytest.mark.parametrize("cfg,cin,cout,rows", [(0, 320, 960, 4096), (1, 1280, 1280, 1024), (2, 1024, 640, 300), (3, 16, 8, 256),
                                               (-1, 2560, 1280, 256), (-1, 640, 640, 4096)])
def test_gemm(eng, cfg, cin, cout, rows):
    S.check_conv(eng, DEV, 1, 1, rows, cin, cout, ntaps=1, tile_cfg=cfg, res="f32", out_f32=True, seed=60 + cfg)


def test_gemm_fp32_input_concat_geglu(eng):
    S.check_conv(eng, DEV, 1, 1, 1000, 320, 320, ntaps=1, in_f32=True, seed=70)                   # proj_out reads the fp32 stream
    S.check_conv(eng, DEV, 1, 8, 8, 1280, 1280, ntaps=1, C1=1280, in_f32=True, res=None, out_f32=True, seed=71)  # shortcut on concat
    S.check_conv(eng, DEV, 1, 1, 1024, 320, 2560, ntaps=1, geglu=True, seed=72)                   # GEGLU
    S.check_conv(eng, DEV, 1, 1, 256, 1280, 10240, ntaps=1, geglu=True, seed=73, atol=8e-3)


def test_groupnorm(eng):
    S.check_groupnorm(eng, DEV, 2, 64, 64, 128, in_f32=True, silu=True)
    S.check_groupnorm(eng, DEV, 1, 32, 32, 320, in_f32=False, silu=False, eps=1e-5)
    S.check_groupnorm(eng, DEV, 2, 8, 8, 1280, C1=1280, in_f32=True, silu=True, eps=1e-5)           # 2560-ch concat
    S.check_groupnorm(eng, DEV, 1, 128, 128, 512, in_f32=True, silu=True)
    S.check_groupnorm(eng, DEV, 3, 7, 5, 64, in_f32=True, silu=True)


def test_layernorm(eng):
    S.check_layernorm(eng, DEV, 4099, 320, in_f32=True)
    S.check_layernorm(eng, DEV, 1024, 1280, in_f32=True)
    S.check_layernorm(eng, DEV, 77, 640, in_f32=False)


def test_attention_d64(eng):
    S.check_attention(eng, DEV, 2, 5, 1024, 1024, 64, use_bias=True, fused_stride=True)
    S.check_attention(eng, DEV, 1, 10, 256, 1024, 64, use_bias=False)                              # cross attention Lq != Lk
    S.check_attention(eng, DEV, 1, 2, 100, 400, 64, use_bias=True, seed=2)                         # ragged (S=640 levels)
    S.check_attention(eng, DEV, 1, 1, 64, 4096, 64, use_bias=False, spike=True, seed=3)            # forced rescale branch


def test_attention_d64_skips_underflowing_key_tiles(eng, engine_option):
    # trimap-like bias with whole key tiles at -5000 / -10000: never loaded; equal to the fp32 reference and bit-identical to
    # walking every tile (option attn_dense)
    S.check_attention(eng, DEV, 3, 5, 1000, 4096, 64, use_bias=True, blocks=True, seed=7)
    import torch
    g = torch.Generator().manual_seed(11)
    q, k, v = (torch.randn(2, n, 128, generator=g).half().to(DEV) for n in (300, 1500, 1500))
    bias = torch.full((2, 1500), -10000.0)
    bias[0, 130:150] = 0.0
    bias[0, 1400:1410] = 0.0
    bias[1, 5:9] = 0.0
    bias[1, 700:] = -5000.0
    sparse = eng.op_attention(q, k, v, 2, bias.to(DEV)).cpu()
    engine_option(eng, "attn_dense", 1)
    dense = eng.op_attention(q, k, v, 2, bias.to(DEV)).cpu()
    assert torch.equal(sparse, dense)


def test_attention_d64_key_split(eng, engine_option):
    """AttnParams::nsplit (engine option attn_ksplit): block row y walks its range of key tiles and leaves unnormalised partial sums, attn_combine_kernel
    merges them.  Against fp64 attention, close to the unsplit kernel, and - the ranges being ranges of KEYS - still bit-identical between the walk
    along the active-tile list and the dense walk; a range without any active tile (an empty part) included."""
    import torch
    for ns in (2, 4):
        engine_option(eng, "attn_ksplit", ns)
        eng.lib.kernel_counts(reset=True)
        S.check_attention(eng, DEV, 2, 5, 1000, 4096, 64, use_bias=True, blocks=True, seed=7, split=True, atol=1e-3)
        S.check_attention(eng, DEV, 1, 2, 700, 1333, 64, use_bias=False, seed=8, split=True, atol=1e-3)
        assert eng.lib.kernel_counts().get("attn_combine", 0) == 2
    g = torch.Generator().manual_seed(12)
    q, k, v = (torch.randn(2, n, 64, generator=g).to(DEV) for n in (300, 6400, 6400))
    bias = torch.full((2, 6400), -10000.0)
    bias[0, 1300:1500] = 0.0          # image 0: active keys in the first quarter only -> three empty parts at nsplit = 4
    bias[1, 5:9] = 0.0
    bias[1, 4000:] = -5000.0
    bias = bias.to(DEV)
    outs = {}
    for ns in (1, 4):
        engine_option(eng, "attn_ksplit", ns)
        engine_option(eng, "attn_dense", 0)
        sparse = eng.op_attention_split(q, k, v, 1, bias).cpu()
        engine_option(eng, "attn_dense", 1)
        dense = eng.op_attention_split(q, k, v, 1, bias).cpu()
        assert torch.equal(sparse, dense), ns
        outs[ns] = sparse
    assert (outs[1] - outs[4]).abs().max().item() < 2e-6


def test_attention_d512(eng):
    S.check_attention(eng, DEV, 1, 1, 1024, 1024, 512, use_bias=False, atol=5e-3)
    S.check_attention(eng, DEV, 2, 1, 200, 320, 512, use_bias=False, atol=5e-3, seed=1)
    S.check_attention(eng, DEV, 2, 1, 130, 301, 512, use_bias=False, atol=5e-3, seed=2)   # ragged last key tile (clamped DMA rows + mask)
    S.check_attention(eng, DEV, 1, 1, 64, 1024, 512, use_bias=False, spike=True, atol=5e-3, seed=3)


def test_resize_aa(eng):
    S.check_resize(eng, DEV, 3, 600, 800, 512, 512)
    S.check_resize(eng, DEV, 1, 512, 512, 600, 800)
    S.check_resize(eng, DEV, 2, 37, 53, 64, 64)


def test_mask_bias_matches_reference_fixture(eng, golden_dir):
    import numpy as np
    g = np.load(os.path.join(golden_dir, "g2_mask_pyramid.npz"))
    tri = torch.from_numpy(g["trimap_m11"])[:, 0].contiguous().to(DEV)          # [B,S,S] in [-1,1]
    for lev, heads in ((0, 5), (1, 10), (2, 20), (3, 20)):
        out = eng.op_mask_bias(tri, lev).cpu()
        ref = torch.from_numpy(g[f"prepared_level{lev}_heads{heads}"])[::heads, 0]   # one row per image
        assert torch.equal(out, ref), f"level {lev}"


def test_attention_matches_reference_scores_fixture(eng, golden_dir):
    import numpy as np
    g = np.load(os.path.join(golden_dir, "g3_attention_scores.npz"))
    q, k, v = (torch.from_numpy(g[n]) for n in ("q", "k", "v"))
    BH, Lq, d = q.shape
    heads = 2
    B = BH // heads
    tok = lambda x: x.view(B, heads, x.shape[1], d).permute(0, 2, 1, 3).reshape(B, x.shape[1], heads * d)
    bias = torch.from_numpy(g["key_bias"])[::heads, 0].contiguous()
    for use_bias, key in ((True, "out_bias"), (False, "out_nobias")):
        out = eng.op_attention(tok(q).half().to(DEV), tok(k).half().to(DEV), tok(v).half().to(DEV), heads, bias.to(DEV) if use_bias else None)
        ref = tok(torch.from_numpy(g[key]))
        # fp16 operands vs the reference's fp32 q/k/v: 5e-3 on O(1) outputs
        assert (out.float().cpu() - ref).abs().max().item() < 5e-3
        # the kernel the DEFAULT precision ships (sdm_op_attention_split: Q.K^T on split operands, P.V on fp16, fp32 output) on the
        # reference's un-rounded fp32 q / k / v: 1e-3 on O(1) outputs (measured 1-5e-4: the fp16 rounding of P and V)
        out2 = eng.op_attention_split(tok(q).to(DEV), tok(k).to(DEV), tok(v).to(DEV), heads, bias.to(DEV) if use_bias else None)
        e2 = (out2.float().cpu() - ref).abs().max().item()
        print(f"[G3 vs shipped split kernel, bias={use_bias}] max|d|={e2:.2e}")
        assert e2 < 1e-3


@pytest.mark.slow
def test_conv3x3_operand_image_beyond_4gb_matches_row_band_crops(eng):
    """One NHWC fp32 image of 256 channels at 2048x2048 is 4.29 GB: more than a buffer descriptor spans.  The kernel's per-tile
    row-band descriptors must give what the same conv (same tile configuration, hence the same arithmetic) gives on small crops
    (rows a-1 .. b+1) of the tensor: 2e-5 on O(10) outputs, where a mis-addressed row would be O(1).
    The crops exercise the first rows, the 4 GB crossing and the last rows."""
    torch.manual_seed(5)
    H = W = 2048
    x = torch.empty(1, H, W, 256, dtype=torch.float32, device="cuda").normal_()
    assert x.numel() * 4 >= (1 << 32)
    w = (torch.randn(128, 256, 3, 3) / 48.0).cuda()
    b = torch.randn(128).cuda()
    res = torch.empty(1, H, W, 128, dtype=torch.float32, device="cuda").normal_()
    for split in (True, False):
        y = eng.op_conv(x, w, b, res=res, out_f32=True, split=split, tile_cfg=0)      # same tile / arithmetic for the image and the crops
        for (ra, rb) in ((0, 16), (2040, 2048), (1016, 1040), (2047 - 16, 2047)):
            lo, hi = max(0, ra - 1), min(H, rb + 1)
            yc = eng.op_conv(x[:, lo:hi].contiguous(), w, b, res=res[:, lo:hi].contiguous(), out_f32=True, split=split, tile_cfg=0)
            got, want = y[:, ra:rb], yc[:, ra - lo:rb - lo]
            tol = 2e-5
            assert (got - want).abs().max().item() <= tol, f"split={split} rows {ra}..{rb}: max|d|={(got - want).abs().max().item():.3e}"
        del y


@pytest.mark.parametrize("cin,cout,H,W,gn,res,up", [(32, 128, 9, 35, None, None, 0), (128, 128, 40, 96, (1e-6, True), "f32", 0), (512, 200, 24, 40, (1e-5, True), None, 0),
                                                   (256, 128, 20, 48, None, "f32", 1), (320, 320, 32, 64, (1e-5, True), "f32", 0)])
def test_conv3x3_fp8_residual_terms(eng, cin, cout, H, W, gn, res, up):
    """F8 kernel (k_conv.h): x_hi.w_hi on fp16 MFMAs, the residual terms x_lo.w and x.w_lo on e4m3 operands through
    v_mfma_scale_f32_32x32x64_f8f6f4, producer / consumer waves, against the un-rounded fp32 reference: far inside the 4e-3 of fp16
    operands, at the ~2^-15 relative level a 2^-4 rounding of a 2^-11 term leaves."""
    err = S.check_conv(eng, DEV, 2, H, W, cin, cout, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, gn=gn, res=res if cout % 4 == 0 else None,
                       up=up, seed=cin + cout, atol=3e-4)
    print(f"[F8 conv {cin}->{cout}] max|d|={err:.2e}")


@pytest.mark.parametrize("xs,ws", [(300.0, 1.0), (1.0, 4.0), (300.0, 4.0), (0.01, 1.0 / 64)])
def test_f8_residual_operand_ranges(eng, xs, ws):
    """Real-checkpoint ranges must not saturate the fp8 residual operands: activations x300 (e5m2 A operands = the range of fp16),
    weights x4 / x1/64 (e4m3 with the layer's own power-of-two scale from max|w|), for the 3x3 and the GEMM form of the F8 kernel:
    relative error stays at the ~2^-14 level of the arithmetic (fixed scales would clamp and fall to ~1e-3)."""
    e = S.check_conv(eng, DEV, 2, 40, 96, 128, 128, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, seed=91, atol=3e-4, xscale=xs, wscale=ws, rel=True)
    g = S.check_conv(eng, DEV, 1, 64, 64, 1280, 320, ntaps=1, tile_cfg=4, in_f32=True, out_f32=True, split=True, f8=True, res="f32", seed=95, atol=3e-4,
                     xscale=xs, wscale=ws, rel=True)
    print(f"[F8 ranges x{xs} w{ws}] conv {e:.2e} gemm {g:.2e}")
    assert e < 1.5e-4 and g < 1.5e-4


@pytest.mark.parametrize("mode", ["0", "3", "4"])
def test_f8_epilogue_modes(eng, engine_option, mode):
    """Option conv_epi: register-direct 16-byte stores + residual as the accumulators' initial value (4, default), residual init alone (3),
    LDS-staged epilogue (0); full and ragged tiles."""
    engine_option(eng, "conv_epi", int(mode))
    S.check_conv(eng, DEV, 2, 64, 128, 128, 160, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, gn=(1e-6, True), res="f32", seed=71, atol=3e-4)
    S.check_conv(eng, DEV, 1, 44, 72, 64, 128, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, res="f32", seed=73, atol=3e-4)
    S.check_conv(eng, DEV, 1, 64, 64, 1280, 320, ntaps=1, tile_cfg=4, in_f32=True, out_f32=True, split=True, f8=True, res="f32", seed=74, atol=3e-4)


def test_attention_d64_split_precision(eng, engine_option):
    """Split-precision attention cores as the engine runs them (Q.K^T on hi | lo pairs, P.V on fp16; 4- and 8-wave blocks), and the
    fully split form, against un-rounded fp64 attention."""
    e2 = S.check_attention(eng, DEV, 2, 5, 300, 1000, 64, use_bias=True, split=True, atol=1e-3)
    engine_option(eng, "attn_nw", 8)
    S.check_attention(eng, DEV, 1, 2, 700, 333, 64, use_bias=False, split=True, seed=5, atol=1e-3)
    engine_option(eng, "attn_nw", 0)
    engine_option(eng, "attn_pv_split", 1)
    e1 = S.check_attention(eng, DEV, 2, 5, 300, 1000, 64, use_bias=True, split=True, atol=3e-5)
    print(f"[split attention] P.V fp16: {e2:.2e}  fully split: {e1:.2e}")


@pytest.mark.parametrize("M_hw,cin,cout,geglu,res", [((64, 64), 320, 320, False, "f32"), ((33, 70), 64, 200, False, None), ((64, 32), 640, 2560, True, None),
                                                     ((128, 128), 256, 128, False, "f32")])
def test_gemm_fp8_residual_terms(eng, M_hw, cin, cout, geglu, res):
    """Linear / 1x1 GEMM on the 8-wave fp8-residual kernel (F8 with NTAPS = 1, the 256 x 128 tile) against the un-rounded fp32
    reference."""
    err = S.check_conv(eng, DEV, 2, M_hw[0], M_hw[1], cin, cout, ntaps=1, tile_cfg=4, in_f32=True, out_f32=True, split=True, f8=True, geglu=geglu,
                       res=res, seed=cin + cout, atol=3e-4)
    print(f"[F8 gemm {cin}->{cout}] max|d|={err:.2e}")


@pytest.mark.parametrize("nw,kern", [(0, "attn_d64_pp"), (8, "attn_d64_pipe<8>"), (4, "attn_d64_pipe<4>")])
def test_attention_pipeline_kernels_with_trimap_bias_and_skipped_tiles(eng, engine_option, golden_dir, nw, kern):
    """The kernels the default precision SHIPS for the d=64 attention cores - the ping-pong of the block's wave halves (round 6: every launch when no
    wave count is forced) and the two-tile software pipelines behind it (options attn_pp = 0 / attn_nw) - against un-rounded fp64 attention with what the engine feeds them: a
    trimap-style key bias whose -10000 / -5000 tiles are skipped (tile lists), ragged query / key counts, >= 5 key tiles; and the
    reference's own scores fixture G3 through the same kernel.  sdm_kernel_counts proves which variant ran."""
    import numpy as np
    engine_option(eng, "attn_nw", nw)
    engine_option(eng, "attn_pp", 1 if nw == 0 else 0)
    engine_option(eng, "attn_pp_min_blocks", 0)      # (the engine keeps launches of < 128 blocks on the pipelines)
    eng.lib.kernel_counts(reset=True)
    lk = (333, 1500, 40) if nw else (320, 1536, 64)      # (the ping-pong kernel takes whole 64-key tiles by LDS-DMA: ragged key counts stay on the pipelines)
    e_blocks = S.check_attention(eng, DEV, 2, 5, 700, lk[0], 64, use_bias=True, split=True, blocks=True, seed=11, atol=1e-3)     # 5-6 key tiles, ragged queries
    e_rand = S.check_attention(eng, DEV, 1, 2, 2100, lk[1], 64, use_bias=True, split=True, seed=12, atol=1e-3)                   # 24 key tiles, random -10000 keys
    e_one = S.check_attention(eng, DEV, 1, 2, 130, lk[2], 64, use_bias=True, split=True, seed=13, atol=1e-3)                     # a single key tile
    e_nobias = S.check_attention(eng, DEV, 2, 5, 300, 1024, 64, use_bias=False, split=True, seed=14, atol=1e-3)                  # no bias (the cross-attention form)
    g = np.load(os.path.join(golden_dir, "g3_attention_scores.npz"))
    q, k, v = (torch.from_numpy(g[n]) for n in ("q", "k", "v"))
    BH, Lq, d = q.shape
    heads = 2
    B = BH // heads
    tok = lambda x: x.view(B, heads, x.shape[1], d).permute(0, 2, 1, 3).reshape(B, x.shape[1], heads * d)
    bias = torch.from_numpy(g["key_bias"])[::heads, 0].contiguous()
    e_g3 = 0.0
    for use_bias, key in ((True, "out_bias"), (False, "out_nobias")):
        out = eng.op_attention_split(tok(q).to(DEV), tok(k).to(DEV), tok(v).to(DEV), heads, bias.to(DEV) if use_bias else None)
        e_g3 = max(e_g3, (out.float().cpu() - tok(torch.from_numpy(g[key]))).abs().max().item())
    counts = eng.lib.kernel_counts()
    print(f"[{kern}] blocks {e_blocks:.2e} random {e_rand:.2e} one tile {e_one:.2e} no bias {e_nobias:.2e} G3 {e_g3:.2e} {counts}")
    assert e_g3 < 1e-3
    if nw:
        assert counts.get(kern, 0) >= 6 and all(c == 0 for n, c in counts.items() if n.startswith("attn_d64") and n != kern), counts
    else:
        assert counts.get(kern, 0) >= 4, counts      # (the fixture's key count may be ragged: those launches fall back to a pipeline)


def test_conv3x3_f8_forced_tiles_per_block(eng, engine_option):
    """Three and four tiles per block (option conv_f8_tpb; the engine picks four from 32 tiles per CU on, i.e. at sizes the CPU reference cannot follow): every
    tile but a block's first is staged by its predecessor - chunk 0 through the previous tile's last steps, chunk 1 in registers across the boundary."""
    for tpb in (3, 4):
        engine_option(eng, "conv_f8_tpb", tpb)
        S.check_conv(eng, DEV, 2, 256, 256, 128, 128, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, gn=(1e-6, True), res="f32", seed=90 + tpb, atol=3e-4)
        S.check_conv(eng, DEV, 1, 128, 256, 256, 256, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, seed=94 + tpb, atol=3e-4)


def test_conv3x3_f8_tiles_back_to_back(eng):
    """F8 3x3 kernel (one A buffer, ring of four weight steps; k_conv.h ConvCfg::R4), 1 / 2 / 4 tiles per block with the cross-tile prefetch: fused
    GroupNorm, residual as accumulator init, odd chunk counts (no prefetch), a concat input."""
    eng.lib.kernel_counts(reset=True)
    S.check_conv(eng, DEV, 2, 64, 128, 128, 128, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, gn=(1e-6, True), res="f32", seed=81, atol=3e-4)
    S.check_conv(eng, DEV, 4, 256, 256, 256, 256, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, gn=(1e-5, True), res=None, seed=82, atol=3e-4)      # 2 tiles per block
    S.check_conv(eng, DEV, 1, 32, 64, 512, 384, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, res="f32", seed=83, atol=3e-4)
    S.check_conv(eng, DEV, 4, 512, 512, 128, 128, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, gn=(1e-6, True), res="f32", seed=84, atol=3e-4)   # 4 tiles per block
    S.check_conv(eng, DEV, 2, 128, 128, 96, 128, C1=64, tile_cfg=0, in_f32=True, out_f32=True, split=True, f8=True, gn=(1e-5, True), seed=85, atol=3e-4)       # 5 chunks
    counts = eng.lib.kernel_counts()
    assert counts.get("conv3x3_f8<gn>", 0) >= 4 and counts.get("conv3x3_f8", 0) >= 1, counts


def test_conv_split_k(eng, engine_option):
    """ConvParams::ksplit + splitk_reduce_kernel on hardware (ops_suite.check_conv_splitk), plus the shapes the engine splits by itself:
    the 16x16 U-Net level at one and four images per call."""
    S.check_conv_splitk(eng, DEV, engine_option)
    eng.lib.kernel_counts(reset=True)
    S.check_conv(eng, DEV, 1, 16, 16, 1280, 1280, in_f32=True, out_f32=True, split=True, res="f32", seed=91, atol=3e-5)
    S.check_conv(eng, DEV, 4, 16, 16, 1280, 640, C1=1280, in_f32=True, out_f32=True, split=True, seed=92, atol=3e-5)
    assert eng.lib.kernel_counts().get("conv3x3_splitk", 0) == 2, eng.lib.kernel_counts()


def test_conv_const_tiles_are_filled_not_multiplied(eng):
    """Piecewise-constant input + class plane through the F8 3x3 kernel: bit-identical to multiplying every tile (ops_suite.check_conv_const_tiles);
    several tiles per block on the larger case (the skipped tiles interrupt the cross-tile prefetch chain)."""
    S.check_conv_const_tiles_are_really_left_out(eng, DEV)
    S.check_conv_const_tiles(eng, DEV)
    S.check_conv_const_tiles(eng, DEV, N=1, H=40, W=96, Cin=64, Cout=128, gn=False, res=False, seed=8)
    S.check_conv_const_tiles(eng, DEV, N=3, H=256, W=512, Cin=128, Cout=256, seed=9)


@pytest.mark.parametrize("tile", [0, 256, 128, 64])
def test_gemm_p3_plane_fed_gemm(eng, engine_option, tile):
    """k_gemm.h on hardware at the transformer blocks' shapes (replace.py:232-362): every epilogue, 10 - 40 K chunks, ragged row / channel tiles,
    image-aligned tiles with fused statistics, LayerNorm with plane output in front; tile 0 = the engine's own choice."""
    so = engine_option
    e0 = S.check_gemm_p3(eng, DEV, 2, 40, 40, 320, 320, mode=0, res=True, tile=tile, seed=1, set_option=so)
    assert e0 < 1.5e-4, e0
    S.check_gemm_p3(eng, DEV, 1, 32, 33, 320, 2560, mode=1, tile=tile, seed=2, set_option=so, atol=1e-4, rel=True)      # u * gelu(g): |values| up to ~15
    S.check_gemm_p3(eng, DEV, 1, 24, 24, 1280, 640, mode=3, res=True, tile=tile, seed=3, set_option=so)
    S.check_gemm_p3(eng, DEV, 2, 17, 19, 640, 1920, mode=2, lo_cols=1280, tile=tile, seed=4, set_option=so)
    S.check_gemm_p3(eng, DEV, 3, 28, 24, 640, 640, mode=4, res=True, tile=tile, seed=5, set_option=so)      # 672 rows per image: ragged image-aligned tiles
    S.check_gemm_p3(eng, DEV, 1, 16, 20, 1280, 1280, mode=0, ln=True, tile=tile, seed=6, set_option=so)
    S.check_gemm_p3(eng, DEV, 1, 7, 9, 64, 96, mode=0, tile=tile, seed=7, set_option=so)


@pytest.mark.parametrize("xs,ws", [(300.0, 4.0), (0.01, 1.0 / 64), (300.0, 1.0 / 64)])
def test_gemm_p3_operand_ranges(eng, xs, ws):
    e = S.check_gemm_p3(eng, DEV, 1, 16, 32, 640, 256, mode=0, seed=11, xscale=xs, wscale=ws, rel=True)
    assert e < 1e-4, (xs, ws, e)
