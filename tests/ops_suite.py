"""Operator parity checks shared by the emulator tests (CPU, tiny shapes) and the GPU tests (real
kernels): every HIP kernel vs a plain torch fp32 reference of the same op on the same seeded inputs.
Tolerances are stated per check: operands are fp16 (MFMA inputs), accumulation fp32."""
import math
import os

import torch
import torch.nn.functional as F


def _g(seed):
    return torch.Generator().manual_seed(seed)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def h16(x):
    """round to fp16 and back: the reference sees exactly the operand values the kernel sees"""
    return x.half().float()


def check_conv(eng, dev, N, H, W, Cin, Cout, ntaps=9, stride=1, pad_mode=0, up=0, C1=0, in_f32=False, res=None, out_f32=False,
               geglu=False, out_scale=1.0, tile_cfg=-1, seed=0, atol=4e-3, split=False, gn=None, f8=False, xscale=1.0, wscale=1.0, rel=False):
    """gn = (eps, silu): GroupNorm(32)(+SiLU) of the input fused into the conv's operand staging (the ResBlock path).
    split = True: split-fp16 operands (precise mode) - the reference then sees the un-rounded fp32 operands.
    f8 = True: the residual terms of the split product on fp8 operands (F8 kernel: 3x3 stride 1, Cin % 32 == 0, tile cfg 0);
    False pins them to fp16 (option conv_f8 = 0) so that the 2e-5 checks of the fp16x3 arithmetic keep their meaning.
    xscale / wscale multiply the N(0,1) activations / the N(0, 1/fan_in) weights (range robustness of the fp8 residual operands);
    rel = True compares max|d| / max|ref| with atol."""
    g = _g(seed)
    rnd = (lambda t: t) if split else h16
    x = torch.randn(N, Cin + C1, H, W, generator=g) * xscale
    x = x if (in_f32 and (split or gn is not None)) else h16(x)
    if ntaps == 9:
        w = torch.randn(Cout, Cin + C1, 3, 3, generator=g) / math.sqrt((Cin + C1) * 9) * wscale
    else:
        w = torch.randn(Cout, Cin + C1, generator=g) / math.sqrt(Cin + C1) * wscale
    b = 0.1 * torch.randn(Cout, generator=g)
    xr = x
    gn_arg = None
    if gn is not None:
        eps, silu = gn
        gamma = 1 + 0.2 * torch.randn(Cin + C1, generator=g)
        beta = 0.1 * torch.randn(Cin + C1, generator=g)
        xr = F.group_norm(x, 32, gamma, beta, eps)
        if silu:
            xr = F.silu(xr)
        xr = rnd(xr)                                   # the normalised value is what gets rounded to the MFMA operand type
        gn_arg = (gamma.to(dev), beta.to(dev), eps, 32, silu)
    if up:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    if ntaps == 9:
        if pad_mode == 1:
            xr = F.pad(xr, (0, 1, 0, 1))
            ref = F.conv2d(xr, rnd(w), b, stride=stride, padding=0)
        else:
            ref = F.conv2d(xr, rnd(w), b, stride=stride, padding=1)
    else:
        ref = F.conv2d(xr, rnd(w)[:, :, None, None], b)
    if geglu:
        u, gg = ref.chunk(2, dim=1)
        ref = u * F.gelu(gg)
    ref = ref * out_scale
    r = None
    if res is not None:
        r = torch.randn(ref.shape, generator=g)
        r = r if res == "f32" else h16(r)
        ref = ref + r
    xs = nhwc(x)
    xs = xs if in_f32 else xs.half()
    x0 = xs[..., :Cin].contiguous().to(dev)
    x1 = xs[..., Cin:].contiguous().to(dev) if C1 else None
    rr = None
    if r is not None:
        rr = nhwc(r)
        rr = (rr if res == "f32" else rr.half()).to(dev)
    prev = eng.lib.get_option("conv_f8")      # (the stand-alone operator packs its weights per call: the option is read there)
    eng.lib.set_option("conv_f8", 1 if f8 else 0)
    try:
        out = eng.op_conv(x0, w.to(dev), b.to(dev), x1=x1, stride=stride, pad_mode=pad_mode, up=up, res=rr, geglu=geglu, out_f32=out_f32,
                          out_scale=out_scale, tile_cfg=tile_cfg, split=split, gn=gn_arg)
    finally:
        eng.lib.set_option("conv_f8", prev)
    got = nchw(out.float().cpu())
    err = (got - ref).abs().max().item()
    if rel:
        err = err / ref.abs().max().item()
    assert err < atol, f"conv mismatch max|d|={err:.4g} (ntaps={ntaps} s={stride} pad={pad_mode} up={up} cfg={tile_cfg} " \
                       f"N={N} H={H} W={W} Cin={Cin}+{C1} Cout={Cout} split={split} gn={gn})"
    return err


def check_conv_splitk(eng, dev, set_option):
    """Split-K of the register-staged conv / GEMM kernels (engine option conv_splitk forced): partial sums + splitk_reduce_kernel against the same
    references and tolerances as the unsplit kernels, for every register-staged tile configuration that takes the split, with concat inputs,
    fp32 / fp16 residuals, fp16 / fp32 outputs and the split-precision operands.  The launch counters prove that the split path ran."""
    cases = [
        # 3x3 stride 1: cfg 2 (64 px x 64 co, KC 16), cfg 1 (128 x 64, KC 32), split precision on both
        dict(N=2, H=16, W=16, Cin=128, Cout=64, tile_cfg=2, ks=4),
        dict(N=1, H=8, W=8, Cin=64, C1=64, Cout=96, tile_cfg=2, ks=2, res="f16"),
        dict(N=1, H=16, W=16, Cin=128, Cout=64, tile_cfg=2, ks=2, in_f32=True, split=True, out_f32=True, res="f32", atol=3e-5),
        dict(N=1, H=8, W=32, Cin=128, Cout=64, tile_cfg=1, ks=2, in_f32=True, split=True, out_f32=True, atol=3e-5),
        dict(N=1, H=8, W=32, Cin=256, Cout=64, tile_cfg=1, ks=4, out_scale=0.5, res="f16"),
        # fused GroupNorm staging (the scale | shift table of the whole input in LDS) with a channel range per block: 256 x 64 and 256 x 128 tiles
        dict(N=2, H=8, W=32, Cin=64, Cout=64, tile_cfg=5, ks=2, gn=(1e-6, True)),
        dict(N=1, H=16, W=32, Cin=128, Cout=96, tile_cfg=0, ks=4, gn=(1e-5, True), in_f32=True, split=True, out_f32=True, res="f32", atol=3e-5),
        # 3x3 stride 2
        dict(N=1, H=16, W=16, Cin=128, Cout=64, stride=2, tile_cfg=1, ks=2),
        # 1x1 / Linear: cfg 2 (64 rows x 64 co, KC 64), cfg 1 (128 x 64), cfg 4 (256 x 128, KC 32: the fp32-input form)
        dict(N=1, H=8, W=8, Cin=512, Cout=128, ntaps=1, tile_cfg=2, ks=4, res="f16"),
        dict(N=2, H=8, W=16, Cin=256, Cout=64, ntaps=1, tile_cfg=1, ks=2, in_f32=True, split=True, out_f32=True, res="f32", atol=3e-5),
        dict(N=1, H=16, W=16, Cin=256, Cout=128, ntaps=1, tile_cfg=4, ks=2, in_f32=True, split=True, out_f32=True, atol=3e-5),
    ]
    for c in cases:
        c = dict(c)
        ks = c.pop("ks")
        set_option(eng, "conv_splitk", ks)
        eng.lib.kernel_counts(reset=True)
        check_conv(eng, dev, **c)
        counts = eng.lib.kernel_counts()
        assert counts.get("conv3x3_splitk", 0) + counts.get("gemm_splitk", 0) == 1, (c, counts)
    set_option(eng, "conv_splitk", -1)


def check_conv_const_tiles(eng, dev, N=2, H=48, W=160, Cin=128, Cout=128, gn=True, res=True, seed=7):
    """The F8 3x3 kernel on a piecewise-constant input with a class plane (what the VAE encoder sees for the trimap images): output tiles inside
    one region are filled from the region's representative tile instead of being multiplied (k_misc.h cmask_*, ConvParams::tile_flag).  The
    result has to be BIT-IDENTICAL to the same launch without the class plane - every output pixel of a region is the same chain of operations
    on the same operands - and the launch counter proves that the skipping path ran."""
    g = _g(seed)
    cls = torch.zeros(N, H, W, dtype=torch.uint8)
    # image 0: class 1 everywhere except a noisy blob in a corner and a class-2 band; image 1: nothing known (an "rgb" image)
    cls[0] = 1
    cls[0, :14, :40] = 0
    cls[0, 30:, 100:] = 2
    vec = torch.randn(3, Cin, generator=g)
    x = torch.randn(N, H, W, Cin, generator=g)
    for c in (1, 2):
        x[cls == c] = vec[c]
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = 0.1 * torch.randn(Cout, generator=g)
    gn_arg = (1 + 0.2 * torch.randn(Cin, generator=g), 0.1 * torch.randn(Cin, generator=g), 1e-6, 32, True) if gn else None
    r = None
    if res:                  # the residual of a ResBlock is constant wherever the block's input was
        rv = torch.randn(3, Cout, generator=g)
        r = torch.randn(N, H, W, Cout, generator=g)
        for c in (1, 2):
            r[cls == c] = rv[c]
    kw = dict(tile_cfg=0, split=True, out_f32=True, gn=tuple(t.to(dev) if torch.is_tensor(t) else t for t in gn_arg) if gn else None,
              res=r.to(dev) if res else None)
    prev = eng.lib.get_option("conv_f8")
    eng.lib.set_option("conv_f8", 1)
    eng.lib.set_option("trimap_skip_min_rows", 8)          # (the engine leaves tiles out from 512 output rows on)
    try:
        full = eng.op_conv(x.to(dev), w.to(dev), b.to(dev), **kw).float().cpu()
        eng.lib.kernel_counts(reset=True)
        skip = eng.op_conv(x.to(dev), w.to(dev), b.to(dev), cmask=cls.to(dev), **kw).float().cpu()
        counts = eng.lib.kernel_counts()
    finally:
        eng.lib.set_option("conv_f8", prev)
        eng.lib.set_option("trimap_skip_min_rows", 512)
    assert counts.get("conv3x3_f8_const_tiles", 0) == 1, counts
    if gn:      # the stand-alone GroupNorm statistics of this test helper are summed with atomics on hardware: two launches differ in the last bits
        assert (full - skip).abs().max().item() <= 2e-5 * full.abs().max().item()
    else:
        assert torch.equal(full, skip), f"constant-tile fill differs from the multiplied tiles: max|d| = {(full - skip).abs().max().item():.3g}"
    # and the region really is constant in the output (away from its border), i.e. there was something to skip
    assert (full[0, 20:28, 48:80] - full[0, 20, 48]).abs().max().item() == 0.0


def check_conv_const_tiles_are_really_left_out(eng, dev, N=2, H=40, W=128, Cin=64, Cout=128, seed=11):
    """The converse of check_conv_const_tiles: a class plane that CLAIMS one constant region over a noise input.  Interior tiles are then filled from
    the representative tile (the first interior tile) and must differ from the multiplied result - proof that the kernel did not multiply them -
    while the border tiles (the zero padding breaks the claim there) and the representative are computed as usual."""
    g = _g(seed)
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = 0.1 * torch.randn(Cout, generator=g)
    cls = torch.ones(N, H, W, dtype=torch.uint8)
    prev = eng.lib.get_option("conv_f8")
    eng.lib.set_option("conv_f8", 1)
    eng.lib.set_option("trimap_skip_min_rows", 8)
    try:
        full = eng.op_conv(x.to(dev), w.to(dev), b.to(dev), tile_cfg=0, split=True, out_f32=True).float().cpu()
        skip = eng.op_conv(x.to(dev), w.to(dev), b.to(dev), tile_cfg=0, split=True, out_f32=True, cmask=cls.to(dev)).float().cpu()
    finally:
        eng.lib.set_option("conv_f8", prev)
        eng.lib.set_option("trimap_skip_min_rows", 512)
    ty, tx = H // 8, W // 32
    differs = ((full - skip).abs().amax(dim=3) > 0).reshape(N, ty, 8, tx, 32).permute(0, 1, 3, 2, 4).reshape(N, ty, tx, 256).any(dim=3)
    expect = torch.zeros(N, ty, tx, dtype=torch.bool)
    expect[:, 1:ty - 1, 1:tx - 1] = True           # interior tiles: every window inside the image
    expect[:, 1, 1] = False                        # the representative of (image, class 1): the smallest interior tile index
    assert torch.equal(differs, expect), (differs, expect)
    for n in range(N):                             # filled tiles hold ONE pixel of the representative, everywhere
        rep = skip[n, 8, 32]
        assert torch.equal(skip[n, 16:24, 32:64], rep.expand(8, 32, -1))


def check_groupnorm(eng, dev, N, H, W, C, C1=0, in_f32=True, silu=True, eps=1e-6, seed=0, atol=4e-3):
    g = _g(seed)
    x = torch.randn(N, C + C1, H, W, generator=g) * 1.7 + 0.3
    if not in_f32:
        x = h16(x)
    gamma = 1 + 0.2 * torch.randn(C + C1, generator=g)
    beta = 0.1 * torch.randn(C + C1, generator=g)
    ref = F.group_norm(x, 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    xs = nhwc(x)
    xs = xs if in_f32 else xs.half()
    x0 = xs[..., :C].contiguous().to(dev)
    x1 = xs[..., C:].contiguous().to(dev) if C1 else None
    out = eng.op_groupnorm(x0, gamma.to(dev), beta.to(dev), eps, silu, x1=x1)
    got = nchw(out.float().cpu())
    err = (got - ref).abs().max().item()
    assert err < atol, f"groupnorm mismatch {err:.4g} (N={N} HW={H}x{W} C={C}+{C1} f32={in_f32} silu={silu})"
    return err


def check_layernorm(eng, dev, rows, C, in_f32=True, seed=0, atol=4e-3):
    g = _g(seed)
    x = torch.randn(rows, C, generator=g) * 2.0 + 0.5
    if not in_f32:
        x = h16(x)
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    ref = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    xs = (x if in_f32 else x.half()).to(dev)
    out = eng.op_layernorm(xs, gamma.to(dev), beta.to(dev), 1e-5)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < atol, f"layernorm mismatch {err:.4g} (rows={rows} C={C})"
    return err


def check_attention(eng, dev, B, heads, Lq, Lk, D, use_bias, seed=0, atol=3e-3, fused_stride=False, spike=False, blocks=False, split=False):
    g = _g(seed)
    q = h16(torch.randn(B, Lq, heads * D, generator=g))
    k = h16(torch.randn(B, Lk, heads * D, generator=g))
    v = h16(torch.randn(B, Lk, heads * D, generator=g))
    if spike:
        # force a large running-max jump late in the key sequence (online-softmax rescale branch)
        k[:, Lk - 3] = h16(q[:, 0] * 4.0)
    bias = None
    if use_bias:
        keep = (torch.rand(B, Lk, generator=g) > 0.5).float()
        keep[:, Lk // 3] = 1.0
        bias = (1 - keep) * -10000.0
        if blocks:
            # trimap-like three-level bias with whole 64-key tiles at -5000 / -10000 (the engine never loads those tiles: their
            # probabilities underflow to exactly 0, as in the reference); image 0: a few scattered foreground runs, last image:
            # no foreground key at all (every key at -10000 -> the bias cancels and nothing may be skipped)
            bias = torch.full((B, Lk), -10000.0)
            bias[:, Lk // 2:] = -5000.0
            for b in range(B - 1 if B > 1 else B):
                for t0 in range(64 * (1 + b), Lk, 64 * 5):
                    bias[b, t0 + 7:min(t0 + 40, Lk)] = 0.0
            if B > 1:
                bias[B - 1] = -10000.0
    scale = D ** -0.5
    qh = q.view(B, Lq, heads, D).permute(0, 2, 1, 3)
    kh = k.view(B, Lk, heads, D).permute(0, 2, 1, 3)
    vh = v.view(B, Lk, heads, D).permute(0, 2, 1, 3)
    s = torch.matmul(qh.double(), kh.double().transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias[:, None, None, :].double()
    ref = torch.matmul(s.softmax(-1), vh.double()).permute(0, 2, 1, 3).reshape(B, Lq, heads * D).float()
    if fused_stride:
        buf = torch.zeros(B, max(Lq, Lk), 3 * heads * D, dtype=torch.float16)
        buf[:, :Lq, :heads * D] = q.half()
        buf[:, :Lk, heads * D:2 * heads * D] = k.half()
        buf[:, :Lk, 2 * heads * D:] = v.half()
        buf = buf.to(dev)
        qd, kd, vd = buf[:, :Lq, :heads * D], buf[:, :Lk, heads * D:2 * heads * D], buf[:, :Lk, 2 * heads * D:]
    else:
        qd, kd, vd = q.half().to(dev), k.half().to(dev), v.half().to(dev)
    if split:
        # the default precision's kernel: Q.K^T on hi | lo operand pairs, P.V on fp16; un-rounded fp32 q / k / v in, fp32 out
        qs, ks, vs = torch.randn(B, Lq, heads * D, generator=g), torch.randn(B, Lk, heads * D, generator=g), torch.randn(B, Lk, heads * D, generator=g)
        s2 = torch.matmul(qs.view(B, Lq, heads, D).permute(0, 2, 1, 3).double(), ks.view(B, Lk, heads, D).permute(0, 2, 1, 3).double().transpose(-1, -2)) * scale
        if bias is not None:
            s2 = s2 + bias[:, None, None, :].double()
        ref = torch.matmul(s2.softmax(-1), vs.view(B, Lk, heads, D).permute(0, 2, 1, 3).double()).permute(0, 2, 1, 3).reshape(B, Lq, heads * D).float()
        out = eng.op_attention_split(qs.to(dev), ks.to(dev), vs.to(dev), heads, bias.to(dev) if bias is not None else None)
    else:
        out = eng.op_attention(qd, kd, vd, heads, bias.to(dev) if bias is not None else None)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < atol, f"attention mismatch {err:.4g} (B={B} h={heads} Lq={Lq} Lk={Lk} D={D} bias={use_bias})"
    return err


def check_resize(eng, dev, P, Hin, Win, Hout, Wout, seed=0, atol=2e-5):
    g = _g(seed)
    x = torch.rand(P, Hin, Win, generator=g)
    ref = F.interpolate(x[None], size=(Hout, Wout), mode="bilinear", align_corners=False, antialias=True)[0]
    out = eng.op_resize_aa(x.to(dev), Hout, Wout)
    err = (out.cpu() - ref).abs().max().item()
    assert err < atol, f"resize mismatch {err:.4g} ({Hin}x{Win}->{Hout}x{Wout})"
    return err


def _e5m2_to_f32(b):
    """uint8 tensor of OCP e5m2 bytes -> float32 (an e5m2 byte is the top byte of an fp16)"""
    return (b.to(torch.int32) << 8).to(torch.int16).view(torch.float16).float()


def check_gemm_p3(eng, dev, N, H, W, K, O, mode=0, res=False, ln=False, lo_cols=-1, tile=0, seed=0, atol=3e-4, xscale=1.0, wscale=1.0, rel=False,
                  set_option=None):
    """Plane-fed GEMM (k_gemm.h): x . w^T (+bias) on pre-split operand planes - fp16 high parts x fp16 high parts + ONE fp8 K=64 MFMA for the two
    residual terms - against the torch fp32 product of the UN-rounded operands.  mode 0 fp32 (+res), 1 GEGLU -> planes, 3 linear (+res) -> planes
    (both decoded: the comparison then includes the 22-bit rounding of the output planes), 2 attention operand planes (hi + pair plane decoded),
    4 fp32 + the consumer's GroupNorm statistics.  ln: LayerNorm with plane output in front of the GEMM.  tile: forced row tile (256 / 128 / 64)."""
    g = _g(seed)
    x = torch.randn(N, H, W, K, generator=g) * xscale
    w = torch.randn(O, K, generator=g) / math.sqrt(K) * wscale
    b = 0.1 * torch.randn(O, generator=g)
    xr = x
    ln_arg = None
    if ln:
        gamma = 1 + 0.2 * torch.randn(K, generator=g)
        beta = 0.1 * torch.randn(K, generator=g)
        xr = F.layer_norm(x, (K,), gamma, beta, 1e-5)
        ln_arg = (gamma.to(dev), beta.to(dev), 1e-5)
    ref = xr.double() @ w.double().t() + b.double()
    if mode == 1:
        u, gg = ref.chunk(2, dim=-1)
        ref = u * F.gelu(gg)
    r = None
    if res:
        r = torch.randn(ref.shape, generator=g)
        ref = ref + r.double()
    ref = ref.float()
    if set_option is not None:
        set_option(eng, "gemm_p3_tile", tile)
    out = eng.op_gemm_p3(x.to(dev), w.to(dev), b.to(dev), mode=mode, res=r.to(dev) if r is not None else None, ln=ln_arg, lo_cols=lo_cols)
    what = f"gemm_p3 mode={mode} N={N} H={H} W={W} K={K} O={O} res={res} ln={ln} tile={tile}"
    if mode == 2:
        hi, pair = out
        hi = hi.float().cpu()
        pair = pair.cpu()                                   # [..., O, 2] bytes: per 4 channels [x8 x 4 | xl x 4]
        p = pair.reshape(N, H, W, O // 4, 8)
        x8 = _e5m2_to_f32(p[..., :4]).reshape(N, H, W, O)
        xl = _e5m2_to_f32(p[..., 4:]).reshape(N, H, W, O)
        lc = O if lo_cols < 0 else lo_cols
        got = hi.clone()
        got[..., :lc] += xl[..., :lc] / 2048.0
        err = (got - ref).abs()
        e_lo = err[..., :lc].max().item() if lc else 0.0
        e_hi = err[..., lc:].max().item() if lc < O else 0.0
        assert e_lo < atol, f"{what}: split planes max|d|={e_lo:.4g}"
        assert e_hi < 2e-3 * max(1.0, ref.abs().max().item()), f"{what}: hi-only channels max|d|={e_hi:.4g}"
        # the e5m2(x) halves of the pair plane: within one e5m2 rounding of the value
        d8 = (x8[..., :lc] - ref[..., :lc]).abs() - 0.126 * ref[..., :lc].abs() - 1e-4
        assert d8.max().item() <= 0, f"{what}: e5m2(x) bytes off by {d8.max().item():.4g}"
        return e_lo
    stats = None
    if mode == 4:
        out, stats = out
    got = out.float().cpu()
    err = (got - ref).abs().max().item()
    if rel:
        err = err / ref.abs().max().item()
    assert err < atol, f"{what}: max|d|={err:.4g}"
    if stats is not None:
        st = stats.cpu().double().sum(dim=1)                # [N, O, 2]
        s1 = got.double().sum(dim=(1, 2)); s2 = (got.double() ** 2).sum(dim=(1, 2))
        assert torch.allclose(st[..., 0], s1, rtol=1e-5, atol=1e-3), f"{what}: statistics (sum)"
        assert torch.allclose(st[..., 1], s2, rtol=1e-5, atol=1e-3), f"{what}: statistics (sum of squares)"
    return err
