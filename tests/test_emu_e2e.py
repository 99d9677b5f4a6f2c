"""The full engine graph (weight packing, arena, every kernel, C ABI) executed on the CPU kernel emulator at tiny
scale and compared with the oracle, plus the world_size-2 data-parallel path over gloo.  The real-kernel versions are the
-m gpu tests; this guards the host logic and the kernel index math in the GPU-less container."""
import ctypes
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _emu_lib():
    from emu.build_emu import build
    from comfyui_sdmatte_amd.engine import Bindings
    return Bindings(ctypes.CDLL(build()))      # (dlopen of one path = one library instance: its options are shared)


def _emu_engine(cfg, precision="fp16"):
    """Emulator engine; the emulator tests that are about host logic / index math run the fp16-operand graph (3x fewer
    emulated MFMAs), test_emu_precise_mode_meets_parity_bar covers the default split-precision graph."""
    from emu.build_emu import build
    from comfyui_sdmatte_amd.engine import Bindings, Engine
    return Engine(cfg, 0, True, _lib=Bindings(ctypes.CDLL(build())), precision=precision)


def test_emu_precise_mode_meets_parity_bar(pkg, engine_option):
    """The default precision ("fp16x3": split-fp16 operands in every conv / GEMM / attention core, fp32 activations) on the kernel
    emulator: alpha within the north star's 1e-3 of the fp32 oracle (the fp16-operand graph sits at ~4e-3 on the same inputs),
    through the generic tiles AND through the 256x128 fused-GroupNorm tile that carries the real sizes."""
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.weights import synthetic_state_dict
    from comfyui_sdmatte_amd.synth import synthetic_inputs
    from comfyui_sdmatte_amd import engine as E
    from oracle import sdmatte_oracle as O
    assert E.PRECISIONS["fp16x3"] == E.PRECISE_ALL and E.precise_mask_of(None) == E.PRECISIONS[E.DEFAULT_PRECISION]
    cfg = SDMatteConfig.tiny()
    w = synthetic_state_dict(cfg, 0)
    img, tri = synthetic_inputs(1, 50, 70, seed=3)
    ref, _ = O.apply_matte(w, cfg.as_dict(), img, tri, 64, mask_refine=False)
    fast = _emu_engine(cfg, "fp16")
    fast.load_state_dict(w)
    dfast = (fast.apply_matte(img, tri, 64) - ref).abs()
    fast.close()
    eng = _emu_engine(cfg, "fp16x3")
    missing, ignored = eng.load_state_dict(w)
    assert missing == [] and ignored == 0
    a = eng.apply_matte(img, tri, 64)
    d = (a - ref).abs()
    assert d.max().item() <= 1e-3 and d.max().item() < 0.25 * dfast.max().item(), (d.max().item(), dfast.max().item())
    engine_option(_emu_lib(), "force_cfg0", 1)
    d0 = (eng.apply_matte(img, tri, 64) - ref).abs()
    assert d0.max().item() <= 1e-3, d0.max().item()
    eng.close()
    # (partial stage masks - the per-stage attribution - run on the GPU: tests/test_gpu_e2e.py::test_e2e_per_stage_precision_attribution)


def test_emu_gpu_node_tail_bit_exact_vs_reference_fixture(pkg, golden_dir):
    """sdm_apply_matte_node = sdm_apply_matte + mask_refine + output composition on the device.  For every
    (output mode, refine, constraint) case of fixture G1, on the fixture's image + trimap: the device tail applied to the engine's
    own alpha == `refine_and_compose` applied to the same alpha on the CPU, bit for bit; `refine_and_compose` itself is pinned
    bit-exactly to what the REFERENCE node produced (tests/test_node_cpu.py::test_cpu_tail_bit_exact_vs_reference)."""
    import numpy as np
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.weights import synthetic_state_dict
    from comfyui_sdmatte_amd.sdmatte_nodes import refine_and_compose
    g = np.load(os.path.join(golden_dir, "g1_node_prepost.npz"))
    image, tri = torch.from_numpy(g["image"]), torch.from_numpy(g["trimap"])
    cfg = SDMatteConfig.tiny()
    eng = _emu_engine(cfg)
    eng.load_state_dict(synthetic_state_dict(cfg, 0))
    # the engine's own resized + clamped alpha for these inputs (per batch size: another batch may select other tiles)
    raws = {2: eng.apply_matte(image, tri, 64), 1: eng.apply_matte(image[:1], tri[:1], 64)}
    # emulator time (~6 s per forward): half of the fixture's cases here - every mode, with and without refine, every constraint
    # value; the GPU suite walks all of them (tests/test_gpu_e2e.py::test_gpu_node_tail_bit_exact_all_fixture_cases)
    for tag in [g["cases"][i] for i in (0, 2, 3, 5, 8, 10)]:
        mode, refine, c = str(tag).split("__")
        nb = 2 if mode == "matted_rgba" else 1                     # emulator time: the whole batch for one mode, one image for the others
        a, m = eng.apply_matte_node(image[:nb], tri[:nb], 64, False, mode, refine == "refine1", int(c[1:]) / 10.0)
        wa, wm = refine_and_compose(raws[nb].clone(), image[:nb], tri[:nb], mode, refine == "refine1", int(c[1:]) / 10.0)   # bit-exact vs G1 (test_node_cpu)
        assert torch.equal(a, wa) and torch.equal(m, wm), str(tag)
    with pytest.raises(ValueError):
        eng.apply_matte_node(image, tri, 64, False, "nope", True, 0.8)
    # a trimap of another size than the image: resized to the inference size on its own, as in the reference (sdmatte_nodes.py:212-214,
    # 349); it only has to match where the reference indexes the alpha with it (mask_refine / matted_rgb), and fails there like it
    from oracle import sdmatte_oracle as O
    tri_small = F.interpolate(tri[:1, None], size=(29, 41), mode="bilinear", align_corners=False)[:, 0].contiguous()
    # ... and is not READ beyond its own size where nothing uses it: the small trimap sits right in front of an inaccessible page
    # (an out-of-bounds load of tri[i], i < H*W of the image, would fault here; on a GPU it can fault the process)
    tri_small = _in_front_of_a_guard_page(tri_small)
    a2, m2 = eng.apply_matte_node(image[:1], tri_small, 64, False, "matted_rgba", False, 0.8)
    # what release_memory() would give back (activation arena + I/O staging) excludes EVERY weight layout, not only the canonical blob:
    # the node's trim threshold (SDMATTE_KEEP_ARENA_GB) is compared with this
    assert 0 < eng.resident_bytes() - eng.weight_bytes() < 2 ** 30 and eng.weight_bytes() >= eng.weight_blob_bytes()
    w = synthetic_state_dict(cfg, 0)
    pred = O.sdmatte_forward(w, cfg.as_dict(), O.preprocess(image[:1], tri_small, 64, False))
    ra, _ = O.postprocess(pred, image[:1], tri_small, "alpha_only", False, 0.8)
    assert a2.shape == image.shape[1:3] or a2.shape == (1,) + tuple(image.shape[1:3])
    assert (a2 - ra).abs().max().item() < 1e-2 and torch.equal(m2[..., :3], image[:1]) and torch.equal(m2[..., 3], a2)
    with pytest.raises(IndexError):
        eng.apply_matte_node(image[:1], tri_small, 64, False, "alpha_only", True, 0.8)       # mask_refine indexes alpha with the trimap
    with pytest.raises(IndexError):
        eng.apply_matte_node(image[:1], tri_small, 64, False, "matted_rgb", False, 0.8)
    with pytest.raises(ValueError):
        eng.apply_matte_node(image[:1, :, :, :2], tri[:1], 64, False, "alpha_only", False, 0.8)    # not an RGB image
    eng.close()


_GUARDED = []      # keeps the mappings alive


def _in_front_of_a_guard_page(t):
    """A copy of the (contiguous, fp32) tensor whose last byte is the last accessible byte of its mapping."""
    import ctypes
    import mmap
    page = mmap.PAGESIZE
    nbytes = t.numel() * t.element_size()
    npages = (nbytes + page - 1) // page
    m = mmap.mmap(-1, (npages + 1) * page)
    base = ctypes.addressof(ctypes.c_char.from_buffer(m))
    libc = ctypes.CDLL(None, use_errno=True)
    assert libc.mprotect(ctypes.c_void_p(base + npages * page), ctypes.c_size_t(page), 0) == 0      # PROT_NONE
    off = npages * page - nbytes
    g = torch.frombuffer(m, dtype=t.dtype, count=t.numel(), offset=off).view(t.shape)
    g.copy_(t)
    _GUARDED.append(m)
    return g


def test_emu_single_process_fan_out(pkg):
    """MultiGpuEngine (the node's single-process multi-GPU path) with two emulator engines standing in for two devices: weights
    packed once and copied, uneven contiguous split, host threads, same bits as one engine."""
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.parallel import MultiGpuEngine
    from comfyui_sdmatte_amd.weights import synthetic_state_dict
    from comfyui_sdmatte_amd.synth import synthetic_inputs
    cfg = SDMatteConfig.tiny()
    w = synthetic_state_dict(cfg, 0)
    one = _emu_engine(cfg)
    one.load_state_dict(w)
    fan = MultiGpuEngine(cfg, [0, 1], _engine_factory=lambda d: _emu_engine(cfg))
    fan.load_state_dict(w)
    img, tri = synthetic_inputs(3, 64, 64, seed=12)
    # same bits as one engine fed the same shards (another batch size may pick other tiles, i.e. another summation order)
    want = torch.cat([one.apply_matte(img[:2], tri[:2], 64), one.apply_matte(img[2:], tri[2:], 64)])
    got = fan.apply_matte(img, tri, 64)
    assert torch.equal(got, want)
    assert (got - one.apply_matte(img, tri, 64)).abs().max().item() < 5e-3
    got1 = fan.apply_matte(img[:1], tri[:1], 64)                     # fewer images than devices
    assert torch.equal(got1, one.apply_matte(img[:1], tri[:1], 64))
    fa, fm = fan.apply_matte_node(img, tri, 64, False, "matted_rgba", True, 0.8)      # the node body (GPU tail) through the fan-out
    oa, om = one.apply_matte_node(img[:2], tri[:2], 64, False, "matted_rgba", True, 0.8)
    assert torch.equal(fa[:2], oa) and torch.equal(fm[:2], om) and fm.shape == (3, 64, 64, 4)
    with pytest.raises(ValueError):                                  # errors propagate from the worker threads
        fan.apply_matte(img, tri[:, :32], 64)
    one.close(); fan.close()


def test_emu_two_host_threads_share_options_and_launch_counters(pkg):
    """Two engines driven by two host threads (what parallel.MultiGpuEngine does with the GIL released inside ctypes) while a third thread flips an
    engine option and reads the launch counters: the counters are a guarded map, and sdm_set_option waits for the forwards in flight (g_opt_mu in
    sdm_engine.cpp), so an option can never change between a forward's arena-sizing pass and its launch pass.  attn_dense selects between two
    bit-identical attention walks, so every result must equal the serial one whenever the flips land."""
    import threading
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.weights import synthetic_state_dict
    from comfyui_sdmatte_amd.synth import synthetic_inputs
    cfg = SDMatteConfig.tiny()
    w = synthetic_state_dict(cfg, 0)
    engs = [_emu_engine(cfg) for _ in range(2)]
    for e in engs:
        e.load_state_dict(w)
    lib = engs[0].lib
    img, tri = synthetic_inputs(2, 64, 64, seed=21)
    want = engs[0].apply_matte(img, tri, 64)
    lib.kernel_counts(reset=True)
    got, errs, stop = [None, None], [], threading.Event()

    def work(i):
        try:
            for _ in range(3):
                got[i] = engs[i].apply_matte(img, tri, 64)
                assert torch.equal(got[i], want)
        except Exception as ex:      # noqa: BLE001
            errs.append(ex)

    def flip():
        v = 0
        while not stop.is_set():
            v ^= 1
            lib.set_option("attn_dense", v)
            lib.kernel_counts()
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    tf = threading.Thread(target=flip)
    for t in ts + [tf]:
        t.start()
    for t in ts:
        t.join()
    stop.set(); tf.join()
    lib.set_option("attn_dense", 0)
    assert not errs, errs
    counts = lib.kernel_counts()
    assert sum(counts.values()) > 0 and all(c > 0 for c in counts.values())
    for e in engs:
        e.close()


def test_emu_full_forward_matches_oracle(pkg):
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.weights import synthetic_state_dict
    from comfyui_sdmatte_amd.synth import synthetic_inputs
    from oracle import sdmatte_oracle as O
    cfg = SDMatteConfig.tiny()
    w = synthetic_state_dict(cfg, 0)
    eng = _emu_engine(cfg)
    missing, ignored = eng.load_state_dict(w)
    assert missing == [] and ignored == 0          # every tensor of the checkpoint schema has a consumer (point_embedding.* too)
    img, tri = synthetic_inputs(1, 64, 64)
    data = O.preprocess(img, tri, 64, False)
    ref = O.sdmatte_forward(w, cfg.as_dict(), data)
    out = eng.forward(data["image"], data["trimap"], is_trans=data["is_trans"].numpy())
    d = (out - ref).abs()
    assert d.max().item() < 1e-2 and d.mean().item() < 1.5e-3, (d.max().item(), d.mean().item())
    # node-level entry with a real resize (50x70 -> 64 -> 50x70)
    img2, tri2 = synthetic_inputs(1, 50, 70)
    a = eng.apply_matte(img2, tri2, 64)
    ra, _ = O.apply_matte(w, cfg.as_dict(), img2, tri2, 64, mask_refine=False)
    d2 = (a - ra).abs()
    assert d2.max().item() < 1e-2 and d2.mean().item() < 1.5e-3
    # the request-stream runner without a process group (world 1) is the same computation
    from comfyui_sdmatte_amd import parallel
    st = parallel.matte_stream(eng, [img2[0]], [tri2[0]], [64], device=torch.device("cpu"))
    assert len(st) == 1 and torch.equal(st[0], a[0])
    # weight blob export/import round trip gives a bit-identical engine
    eng2 = _emu_engine(cfg)
    blob = torch.empty(eng.weight_blob_bytes(), dtype=torch.uint8)
    hblob = torch.empty(eng.host_blob_bytes(), dtype=torch.uint8)
    eng.export_weights(blob, hblob)
    eng2.import_weights(blob, hblob)
    out2 = eng2.forward(data["image"], data["trimap"], is_trans=data["is_trans"].numpy())
    assert torch.equal(out, out2)
    # shape mismatch must raise like torch's load_state_dict
    bad = dict(w)
    bad["unet.conv_in.weight"] = torch.zeros(3, 3, 3, 3)
    with pytest.raises(RuntimeError):
        eng2.load_state_dict(bad)
    eng.close()
    eng2.close()


def test_emu_fused_groupnorm_path(pkg, engine_option):
    """Force the 256x128 conv tile so that GroupNorm+SiLU is applied inside the conv operand staging (the production path at
    real sizes) and compare with the oracle and with the un-fused path."""
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.weights import synthetic_state_dict
    from comfyui_sdmatte_amd.synth import synthetic_inputs
    from oracle import sdmatte_oracle as O
    cfg = SDMatteConfig.tiny()
    w = synthetic_state_dict(cfg, 0)
    img, tri = synthetic_inputs(1, 64, 64)
    ref, _ = O.apply_matte(w, cfg.as_dict(), img, tri, 64, mask_refine=False)
    eng = _emu_engine(cfg)
    eng.load_state_dict(w)
    engine_option(_emu_lib(), "force_cfg0", 1)
    eng.profile(True)
    a = eng.apply_matte(img, tri, 64)
    d = (a - ref).abs()
    assert d.max().item() < 1e-2 and d.mean().item() < 1.5e-3, (d.max().item(), d.mean().item())
    eng.close()


def _stream_requests():
    """seven requests, two inference sizes, four image shapes: with two ranks the remote rank returns at least three alphas of unequal shapes in its one
    packed message"""
    from comfyui_sdmatte_amd.synth import synthetic_inputs
    i0, t0 = synthetic_inputs(2, 64, 64, seed=21)
    i1, t1 = synthetic_inputs(2, 50, 70, seed=22)
    i2, t2 = synthetic_inputs(2, 40, 56, seed=23)
    i3, t3 = synthetic_inputs(1, 72, 48, seed=24)
    return [(i0[0], t0[0], 64), (i1[0], t1[0], 128), (i0[1], t0[1], 64), (i2[0], t2[0], 64), (i1[1], t1[1], 128), (i3[0], t3[0], 64), (i2[1], t2[1], 64)]


def _dp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    load_package()
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.weights import synthetic_state_dict
    from comfyui_sdmatte_amd.synth import synthetic_inputs
    from comfyui_sdmatte_amd import parallel
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = SDMatteConfig.tiny()
    eng = _emu_engine(cfg, "fp16x3")          # the default precision: the two-rank results are held to the north star's 1e-3 below
    if rank == 0:
        eng.load_state_dict(synthetic_state_dict(cfg, 0))
    parallel.broadcast_weights(eng, 0, torch.device("cpu"))
    img, tri = synthetic_inputs(2, 64, 64)                       # global batch of 2, one image per rank
    lo, hi = parallel.shard_range(2, world, rank)
    a = eng.apply_matte(img[lo:hi], tri[lo:hi], 64)
    outs = parallel.gather_alphas(a, 0)
    # mixed-resolution request stream (BASELINE config #5 at toy scale): 3 requests, two inference sizes, two image shapes
    reqs = _stream_requests()
    res = parallel.matte_stream(eng, [r[0] for r in reqs], [r[1] for r in reqs], [r[2] for r in reqs], micro_batch=2, dst=0,
                                device=torch.device("cpu"))
    if rank == 0:
        q.put((torch.cat(outs, 0), res))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def test_data_parallel_two_ranks_gloo(pkg):
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.weights import synthetic_state_dict
    from comfyui_sdmatte_amd.synth import synthetic_inputs
    from comfyui_sdmatte_amd import parallel
    from oracle import sdmatte_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, stream = q.get(timeout=900)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    cfg = SDMatteConfig.tiny()
    w = synthetic_state_dict(cfg, 0)
    img, tri = synthetic_inputs(2, 64, 64)
    ref, _ = O.apply_matte(w, cfg.as_dict(), img, tri, 64, mask_refine=False)
    d = (got - ref).abs()
    assert got.shape == ref.shape and d.max().item() <= 1e-3, (d.max().item(), d.mean().item())
    # the request stream: every alpha back on rank 0, in request order, at its own resolution
    reqs = _stream_requests()
    assert len(stream) == len(reqs)
    for (im, tr, S), a in zip(reqs, stream):
        r, _ = O.apply_matte(w, cfg.as_dict(), im[None], tr[None], S, mask_refine=False)
        dd = (a - r[0]).abs()
        assert a.shape == im.shape[:2] and dd.max().item() <= 1e-3, (S, dd.max().item(), dd.mean().item())
    # partitioning helpers
    assert [parallel.shard_range(32, 8, r) for r in (0, 7)] == [(0, 4), (28, 32)]
    assert parallel.shard_range(5, 4, 3) == (5, 5)
    plan = parallel.bucket_requests([512, 768, 1024] * 8, 4)
    assert sorted(i for p in plan for v in p.values() for i in v) == list(range(24))
    loads = [sum(parallel.FLOPS_PER_IMAGE[s] * len(v) for s, v in p.items()) for p in plan]
    assert max(loads) / min(loads) < 1.35
    # the stream above really exercised a packed multi-alpha message: the remote rank owned >= 3 requests of unequal shapes
    remote = sorted(i for v in parallel.bucket_requests([r[2] for r in reqs], 2)[1].values() for i in v)
    assert len(remote) >= 3 and len({tuple(reqs[i][0].shape[:2]) for i in remote}) >= 2


def test_emu_other_prompt_types_match_oracle(pkg):
    """The other prompt types of the reference core (meta_arch.py:22-28,131-206) through the core API on
    the emulator: box prompt with per-image coordinates, mask prompt without the key mask, point prompts (point_embedding,
    padded coordinate count incl. an odd channel count), use_coor_input=False."""
    import dataclasses
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.core import SDMatte
    from comfyui_sdmatte_amd.weights import synthetic_state_dict
    from comfyui_sdmatte_amd.synth import synthetic_inputs
    from oracle import sdmatte_oracle as O

    def run(cfg, aux_input, data, **kw):
        w = synthetic_state_dict(cfg, 3)
        m = SDMatte(None, use_aux_input=True, aux_input=aux_input, load_weight=False, config=cfg, **kw)
        m.load_state_dict(w, strict=False)
        m.engine = _emu_engine(cfg)
        m._upload()
        out = m(data)
        okw = {k: v for k, v in kw.items() if k in ("use_coor_input", "use_attention_mask", "attn_mask_aux_input")}
        ref = O.sdmatte_forward(w, cfg.as_dict(), data, aux_input=aux_input, **okw)
        d = (out - ref).abs()
        m.engine.close()
        assert d.max().item() < 1e-2 and d.mean().item() < 1.5e-3, (aux_input, kw, d.max().item(), d.mean().item())
        return out

    cfg = SDMatteConfig.tiny()
    img, tri = synthetic_inputs(2, 64, 64, seed=5)
    base = O.preprocess(img, tri, 64, False)
    g = torch.Generator().manual_seed(9)
    # box prompt: two different boxes in one batch (two conditioning variants), key mask from the box mask
    data = {"image": base["image"], "is_trans": torch.tensor([0, 1]), "bbox_mask": base["trimap"],
            "bbox_coords": torch.tensor([[0.1, 0.2, 0.7, 0.9], [0.0, 0.3, 0.5, 1.0]])}
    a = run(cfg, "bbox_mask", data, attn_mask_aux_input=("point_mask", "bbox_mask", "mask"))
    # the same inputs with the default box (use_coor_input=False) must differ: the coordinates really reach the network
    b = run(cfg, "bbox_mask", data, attn_mask_aux_input=("point_mask", "bbox_mask", "mask"), use_coor_input=False)
    assert (a - b).abs().max().item() > 1e-4
    # mask prompt outside attn_mask_aux_input: no key mask
    data = {"image": base["image"], "is_trans": torch.tensor([0, 0]), "mask": base["trimap"],
            "mask_coords": torch.tensor([[0.0, 0.0, 1.0, 1.0]] * 2)}
    run(cfg, "mask", data, attn_mask_aux_input=("point_mask", "bbox_mask"))
    # point prompts: 5 coordinates -> padded to 8 x 8 channels (P = 64); zeroed coordinates with use_coor_input=False
    data = {"image": base["image"], "is_trans": torch.tensor([1, 0]), "point_mask": base["trimap"],
            "point_coords": torch.rand(2, 5, generator=g)}
    c = run(cfg, "point_mask", data, attn_mask_aux_input=("point_mask", "bbox_mask", "mask"))
    d0 = run(cfg, "point_mask", data, attn_mask_aux_input=("point_mask", "bbox_mask", "mask"), use_coor_input=False)
    assert (c - d0).abs().max().item() > 1e-4
    # P = 60: 11 coordinates -> padded to 12 x 5 channels (odd channel count: one zero column per coordinate)
    cfg60 = dataclasses.replace(cfg, point_embeddings_input_dim=60)
    data["point_coords"] = torch.rand(2, 11, generator=g)
    run(cfg60, "point_mask", data, attn_mask_aux_input=("point_mask", "bbox_mask", "mask"))
    # coordinates that cannot be padded to a divisor (N >= P) fail loudly, as the reference's loop would
    data["point_coords"] = torch.rand(2, 64, generator=g)
    with pytest.raises(RuntimeError):
        run(cfg, "point_mask", data)
    # unsupported constructor options still raise
    with pytest.raises(NotImplementedError):
        SDMatte(None, use_aux_input=True, aux_input=None, load_weight=False, config=cfg)
    with pytest.raises(NotImplementedError):
        SDMatte(None, use_aux_input=True, aux_input="trimap", add_noise=True, load_weight=False, config=cfg)


def test_emu_rectangular_inference_matches_oracle(pkg):
    """Rectangular inference (an extension: the reference asserts square latents): core API with
    [B,3,64,128] inputs vs the oracle's rectangular restatement, incl. the 4-level key-bias pyramid on a 8x16 token grid."""
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.weights import synthetic_state_dict
    from comfyui_sdmatte_amd.synth import synthetic_inputs
    from oracle import sdmatte_oracle as O
    cfg = SDMatteConfig.tiny()
    w = synthetic_state_dict(cfg, 4)
    eng = _emu_engine(cfg)
    eng.load_state_dict(w)
    img, tri = synthetic_inputs(1, 64, 128, seed=8)
    data = {"image": (img.permute(0, 3, 1, 2).contiguous() - 0.5) / 0.5, "trimap": tri.unsqueeze(1) * 2 - 1,
            "is_trans": torch.tensor([0]), "trimap_coords": torch.tensor([[0.0, 0.0, 1.0, 1.0]])}
    ref = O.sdmatte_forward(w, cfg.as_dict(), data)
    out = eng.forward(data["image"], data["trimap"], is_trans=data["is_trans"].numpy())
    d = (out - ref).abs()
    assert out.shape == (1, 1, 64, 128) and d.max().item() < 1e-2 and d.mean().item() < 1.5e-3, (d.max().item(), d.mean().item())
    with pytest.raises(RuntimeError):
        eng.forward(data["image"][..., :96], data["trimap"][..., :96])          # 96 is not a multiple of 64
    eng.close()


def test_bench_two_ranks_control_flow(pkg):
    """`bench.py --gpus 2` has never met a second GPU in the build container: its launcher re-exec, rank / world bookkeeping, weight
    broadcast, per-step gather, max-over-ranks timing and rank-0 JSON line are exercised here on the kernel emulator over gloo
    (`--emu`: tiny architecture, 64x64, one step - a control-flow test, not a measurement)."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emu", "--gpus", "2", "--size", "64", "--batch", "1", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]              # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["steps"] == 1
    assert d["config"]["global_batch"] == 2 and d["config"]["parallelism"] == "dp2"
    assert d["ranks_seen"] == 2 and len(d["per_rank_images_per_s"]) == 2      # the collective really spanned both ranks
    assert d["cpu_baseline"] is None
    # the configs[4] leg (--stream): FLOP-balanced buckets per rank, matted_rgba through the node body, one packed message per peer
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emu", "--gpus", "2", "--size", "64", "--batch", "2", "--steps", "1", "--warmup", "0",
                        "--stream", "--stream-requests", "3"], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "stream" in d["metric"] and "configs[4]" in d["config"]["workload"] and d["ranks_seen"] == 2


def test_checkpoint_variants_load_to_identical_engines(pkg, tmp_path):
    """What a real `SDMatte*.safetensors` can look like without us ever having seen one (sdmatte_nodes.py:298-323 loads it with
    strict=False): saved in fp16 or bf16, the VAE attention under diffusers' legacy names (query / key / value / proj_attn), a
    text_encoder.* subtree the trimap path never touches.  Each variant, streamed through the node's loader (LazyCheckpoint ->
    sdm_load_tensor), must give bit-identical packed weights to an fp32 file with the modern names holding the same values; and
    tools/check_checkpoint.py must pass the same files from their headers alone."""
    import json
    import subprocess
    import sys
    from safetensors.torch import save_file
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.weights import synthetic_state_dict
    from comfyui_sdmatte_amd.sdmatte_nodes import LazyCheckpoint
    cfg = SDMatteConfig.tiny()
    w = synthetic_state_dict(cfg, 3)

    def blob_of(path):
        eng = _emu_engine(cfg, precision="fp16x3")
        missing, ignored = eng.load_state_dict(LazyCheckpoint(str(path)), strict=True)
        dev = torch.empty(eng.weight_blob_bytes(), dtype=torch.uint8)
        host = torch.empty(eng.host_blob_bytes(), dtype=torch.uint8)
        eng.export_weights(dev, host)
        eng.close()
        return dev, host, ignored

    def legacy_names(sd):
        out = {}
        for k, v in sd.items():
            if k.startswith("vae.") and "mid_block.attentions.0" in k:
                for a, b in ((".to_q.", ".query."), (".to_k.", ".key."), (".to_v.", ".value."), (".to_out.0.", ".proj_attn.")):
                    k = k.replace(a, b)
            out[k] = v
        return out

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "check_checkpoint.py")
    for dt in (torch.float16, torch.bfloat16):
        rounded = {k: v.to(dt).float().contiguous() for k, v in w.items()}                       # the values such a file holds
        f32 = tmp_path / f"f32_{dt}.safetensors"
        save_file(rounded, str(f32))
        low = {k: v.to(dt).contiguous() for k, v in legacy_names(w).items()}
        low["text_encoder.text_model.embeddings.token_embedding.weight"] = torch.zeros(8, 4, dtype=dt)
        low["text_encoder.text_model.final_layer_norm.bias"] = torch.zeros(4, dtype=dt)
        lowf = tmp_path / f"low_{dt}.safetensors"
        save_file(low, str(lowf))
        d0, h0, ig0 = blob_of(f32)
        d1, h1, ig1 = blob_of(lowf)
        assert torch.equal(d0, d1) and torch.equal(h0, h1), str(dt)
        assert ig0 == 0 and ig1 == 0                                                              # text_encoder.* never reaches the engine
        r = subprocess.run([sys.executable, tool, str(lowf), "--config", "tiny", "--write-manifest", str(tmp_path / "m.json")], capture_output=True, text=True)
        assert r.returncode == 0 and "missing (engine would refuse to load): 0" in r.stdout and "shape mismatches: 0" in r.stdout, r.stdout + r.stderr
        r = subprocess.run([sys.executable, tool, "--diff", str(tmp_path / "m.json"), "--config", "tiny"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([sys.executable, tool, "--write-expected", str(tmp_path / "exp.json"), "--config", "full"], capture_output=True, text=True)
    assert r.returncode == 0
    exp = json.load(open(tmp_path / "exp.json"))["tensors"]
    assert exp["unet.conv_in.weight"]["shape"] == [320, 8, 3, 3] and exp["unet.aux_conv_in.weight"]["shape"] == [1024, 4, 3, 3] and len(exp) > 900
