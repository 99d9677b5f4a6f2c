#!/usr/bin/env python3
"""bench.py - alpha mattes/sec at 1024x1024 on N MI355X (BASELINE.json metric), one process per GPU.

  python bench.py --gpus N --steps K --warmup W       (N > 1 without a launcher: re-executes itself under torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path (resize-free preprocessing -> VAE encode x2 -> U-Net -> VAE decode -> alpha) over a
batch of B synthetic 1024x1024 image+trimap pairs per GPU that are already resident in HBM, through the same C-ABI entry
point the ComfyUI node uses, followed by the gather of the alphas to rank 0 (RCCL).  Weights: the real SD-2.1/SDMatte
architecture with seed-fixed synthetic weights (no checkpoint and no network on the box); rank 0 packs them once and
broadcasts the packed blob to the other ranks over RCCL.  Scaling is weak (B images per GPU, independent images, no
data-path collective other than the alpha gather).

`value` is measured in the engine's DEFAULT precision ("fp16x3": split-fp16 MFMA operands + fp32 activations, the mode that
meets the north star's 1e-3 alpha tolerance against the fp32 reference path; `parity` in the JSON line is measured in the
same process).  The opt-in fast mode (plain fp16 operands, ~4e-3 from the reference) is timed next to it under `modes`.

Prints ONE JSON line on rank 0 (see the driver contract) including
  "roofline"     : achieved vs peak for the dominant kernel (conv3x3 implicit-GEMM MFMA), HIP-event timed per launch
  "cpu_baseline" : the fp32 CPU oracle (port of the reference's force_cpu path) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

FLOPS_PER_IMAGE = {512: 5.96e12, 768: 14.59e12, 1024: 28.89e12}     # SURVEY.md 8d (dense, 2*MAC)
MFMA_F16_PEAK_TFLOPS = 2500.0                                       # MI355X dense fp16 (MI355X_MICROARCH.md)


def timed_steps(step, steps, world, dev):
    """K steps bracketed by barrier + synchronize on both sides; returns max-over-ranks seconds."""
    import torch.distributed as dist
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)      # (--emu test hook: host memory, nothing to wait for)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    PER_RANK_S[:] = [elapsed]
    if world > 1:
        # every rank's own time (all_gather: the list's length is the number of ranks the collective REALLY spanned), then the maximum
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        got = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(got, t)
        PER_RANK_S[:] = [float(x.item()) for x in got]
        elapsed = max(PER_RANK_S)
    return elapsed


PER_RANK_S = []      # seconds of the last timed_steps() per rank (filled on every rank)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--precision", type=str, default=None, help="fp16x3 (default: meets the 1e-3 parity bar) or fp16 (fast)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-mode", action="store_true", help="skip timing the non-default precision")
    ap.add_argument("--timed-only", action="store_true",
                    help="only warmup + timed steps + the one profiled step (all identical): the run rocprofv3 summaries are taken from")
    ap.add_argument("--cpu-sample-size", type=int, default=1024,
                    help="size of the ONE image the fp32 CPU oracle is timed on (default: the benchmark's own 1024 - about two minutes of host time)")
    ap.add_argument("--emu", action="store_true",
                    help="TEST HOOK (tests/test_emu_e2e.py): run the same script on the CPU kernel emulator with gloo and the tiny architecture, so "
                         "that the N > 1 control flow of this file is exercised in the GPU-less build container; never a benchmark")
    ap.add_argument("--dump-profile", type=str, default=None, help="write the per-launch profile CSV here")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="engine kernel-selection option for A/B runs (sdm_set_option; see sdm_option_name / sdm_option_help)")
    ap.add_argument("--stream", action="store_true",
                    help="BASELINE configs[4] instead of configs[1]: a step = one mixed-resolution request stream (inference sizes 512 / 768 / 1024 in turn, "
                         "--stream-requests per GPU, matted_rgba output) through parallel.matte_stream - FLOP-balanced buckets per rank, one packed "
                         "message per peer back to rank 0; same JSON line, metric and workload named accordingly")
    ap.add_argument("--stream-requests", type=int, default=6, help="requests per GPU and step of the --stream leg")
    ap.add_argument("--dense-attention", action="store_true",
                    help="walk every key tile in the trimap-biased self-attention instead of skipping the tiles whose bias underflows the softmax")
    args = ap.parse_args()
    if args.timed_only:
        args.no_other_mode = args.no_cpu_baseline = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, RCCL rendezvous on 127.0.0.1)
        if not args.emu and torch.cuda.device_count() < args.gpus:
            sys.exit(f"[bench] --gpus {args.gpus} requested but only {torch.cuda.device_count()} GPU(s) are visible")
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"[bench] WORLD_SIZE={world} does not match --gpus {args.gpus}")
    import torch.distributed as dist
    if args.emu:
        dev = torch.device("cpu")
        args.timed_only = args.no_other_mode = args.no_cpu_baseline = True
        torch.set_num_threads(2)
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.emu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    load_package()
    from comfyui_sdmatte_amd import engine as E
    from comfyui_sdmatte_amd.config import SDMatteConfig
    from comfyui_sdmatte_amd.synth import synthetic_inputs
    from comfyui_sdmatte_amd.weights import synthetic_state_dict

    cfg = SDMatteConfig.tiny() if args.emu else SDMatteConfig.full()
    S, B = args.size, args.batch
    precision = args.precision or E.DEFAULT_PRECISION
    other = "fp16" if precision != "fp16" else "fp16x3"
    if args.emu:
        import ctypes
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from emu.build_emu import build as _build_emu
        emu_lib = E.Bindings(ctypes.CDLL(_build_emu()))
        for kv in args.opt:
            emu_lib.set_option(kv.split("=", 1)[0], int(kv.split("=", 1)[1]))
        eng = E.Engine(cfg, 0, precision=precision, _lib=emu_lib)
    else:
        for kv in args.opt:                      # (before the model is built: some options select the weight layouts)
            E.load_library().set_option(kv.split("=", 1)[0], int(kv.split("=", 1)[1]))
        eng = E.Engine(cfg, local_rank, precision=precision)
    if args.dense_attention:
        eng.lib.set_option("attn_dense", 1)      # (an engine option set through the C ABI: the library reads no environment variable)
    sd = None
    if rank == 0:
        sd = synthetic_state_dict(cfg, 0)
    t_load0 = time.time()
    if rank == 0:
        missing, _ = eng.load_state_dict(sd)          # fp32 host tensors -> pinned staging ring -> HIP pack kernels (hi | lo planes)
        assert not missing, missing[:4]
    if world > 1:
        # RCCL broadcast of the packed weight blob (+ the small host-side embedding tensors at its tail)
        from comfyui_sdmatte_amd.parallel import broadcast_weights
        broadcast_weights(eng, 0, dev)
        if dev.type == "cuda":
            torch.cuda.empty_cache()
    load_s = time.time() - t_load0

    # synthetic inputs, resident in HBM (H = W = S so the node's resizes are identities)
    img, tri = synthetic_inputs(B, S, S, seed=1234 + rank)
    img_d, tri_d = img.to(dev), tri.to(dev)
    alpha = torch.empty(B, S, S, dtype=torch.float32, device=dev)
    gathered = [torch.empty_like(alpha) for _ in range(world)] if (world > 1 and rank == 0) else None

    def make_step(engine):
        def step():
            engine.apply_matte(img_d, tri_d, S, False, out=alpha, sync=dev.type != "cuda")    # stream-ordered with torch's current stream
            if world > 1:
                dist.gather(alpha, gathered, dst=0)                              # RCCL on the same stream order: no host sync in between
        return step

    step = make_step(eng)
    stream_sizes = None
    if args.stream:
        # configs[4]: every rank holds the whole request list (resident in its HBM), runs the buckets `bucket_requests` gives it, rank 0 receives
        from comfyui_sdmatte_amd.parallel import matte_stream
        cyc = (64, 128, 64) if args.emu else (512, 768, 1024)
        stream_sizes = [cyc[i % 3] for i in range(args.stream_requests * world)]
        reqs = []
        for i, s_ in enumerate(stream_sizes):
            im, tr = synthetic_inputs(1, s_, s_, seed=4321 + i)
            reqs.append((im[0].to(dev), tr[0].to(dev)))

        def step():      # noqa: F811 - the stream leg replaces the uniform-batch step
            return matte_stream(eng, [r[0] for r in reqs], [r[1] for r in reqs], stream_sizes, micro_batch=B, dst=0, device=dev, output_mode="matted_rgba")
    for _ in range(args.warmup):
        step()
    elapsed = timed_steps(step, args.steps, world, dev)
    per_rank_s = list(PER_RANK_S)
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    value = world * B * args.steps / elapsed
    if args.stream:
        value = len(stream_sizes) * args.steps / elapsed
        args.timed_only = True      # (the companion legs below describe the uniform 1024^2 batch)

    if rank == 0:
        # ---- B = 1 latency (BASELINE configs[1] is quoted one image per call) ----
        b1 = None
        if world == 1 and not args.timed_only:
            a1 = torch.empty(1, S, S, dtype=torch.float32, device=dev)
            i1, t1 = img_d[:1].contiguous(), tri_d[:1].contiguous()
            eng.apply_matte(i1, t1, S, False, out=a1, sync=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n1 = 3
            for _ in range(n1):
                eng.apply_matte(i1, t1, S, False, out=a1, sync=False)
            torch.cuda.synchronize()
            ms1 = (time.perf_counter() - t0) * 1e3 / n1
            b1 = {"batch": 1, "ms_per_image": round(ms1, 3), "images_per_s": round(1e3 / ms1, 3)}
        # ---- host-buffer hand-over (PCIe-inclusive; never `value`): pageable host tensors in, host alpha out ----
        incl = None
        if world == 1 and not args.timed_only:
            ah = torch.empty(B, S, S, dtype=torch.float32)
            eng.apply_matte(img, tri, S, False, out=ah)
            t0 = time.perf_counter()
            for _ in range(2):
                eng.apply_matte(img, tri, S, False, out=ah)
            dt = (time.perf_counter() - t0) / 2
            incl = {"images_per_s": round(B / dt, 3), "ms_per_step": round(dt * 1e3, 2),
                    "note": "inputs handed over as pageable host buffers: 16 MB H2D + 4 MB D2H per image inside the timed region"}
        # ---- the same step with the self-attention walking EVERY key tile (the headline skips the tiles whose trimap bias underflows
        #      the fp32 softmax - exact, but trimap-dependent): reported next to `value`, never instead of it ----
        def companion(opts, note):
            """the timed step again with some engine options changed; the previous values (possibly the user's --opt) are restored afterwards"""
            prev = {k: eng.lib.get_option(k) for k in opts}
            for k, v in opts.items():
                eng.lib.set_option(k, v)
            try:
                step()
                nd = max(2, args.steps // 2)
                el_d = timed_steps(step, nd, 1, dev)
                return {"images_per_s": round(B * nd / el_d, 3), "ms_per_step": round(el_d * 1e3 / nd, 3), "note": note}
            finally:
                for k, v in prev.items():
                    eng.lib.set_option(k, v)
        dense = None
        if world == 1 and not args.timed_only and eng.lib.get_option("attn_dense") == 0:
            dense = companion({"attn_dense": 1}, "engine option attn_dense = 1: every key tile of the trimap-biased self-attention is loaded and multiplied")
        # ---- the same step with every tile of the VAE encoder's trimap images multiplied (the headline fills the output tiles that lie inside
        #      a constant region of the trimap from one representative tile - exact, but trimap-dependent): reported next to `value` ----
        all_tiles = None
        if world == 1 and not args.timed_only and eng.lib.get_option("trimap_skip") != 0:
            all_tiles = companion({"trimap_skip": 0}, "engine option trimap_skip = 0: every output tile of the encoder convs is multiplied, whatever the trimap")
        # ---- neither of the two input-dependent shortcuts: the trimap-independent rate ----
        no_short = None
        if world == 1 and not args.timed_only and (eng.lib.get_option("trimap_skip") != 0 or eng.lib.get_option("attn_dense") == 0):
            no_short = companion({"trimap_skip": 0, "attn_dense": 1},
                                 "attn_dense = 1 and trimap_skip = 0 together: every key tile loaded, every conv tile multiplied - what the step costs on ANY trimap")
        # ---- roofline of the dominant kernel: per-launch HIP events on the engine stream (separate passes, 1 step each).  The roofline is taken
        #      with trimap_skip = 0: every counted FLOP of the conv family is then a multiplied one (utilisation from EXECUTED
        #      flops); the pass with the engine's defaults is what kernel_breakdown_ms describes and gives the dense-equivalent figure beside it ----
        def profile_pass():
            eng.profile(True)
            eng.apply_matte(img_d, tri_d, S, False, out=alpha, sync=True)
            eng.profile(False)
            return eng.profile_results()
        prof = profile_pass()
        prof_dump = eng.profile_dump() if args.dump_profile else None
        prof_exec = prof
        skip_prev = eng.lib.get_option("trimap_skip")
        if skip_prev != 0 and not args.timed_only:       # (--timed-only: every step of the run stays identical, for the rocprofv3 summaries)
            eng.lib.set_option("trimap_skip", 0)
            try:
                prof_exec = profile_pass()
            finally:
                eng.lib.set_option("trimap_skip", skip_prev)
        if args.dump_profile:
            with open(args.dump_profile, "w") as fh:
                fh.write(prof_dump)
        roof = None
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
        # (profiles/pmc_traffic.json, written by tools/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE passes, KB units,
        # FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md); null when no matching profile is committed.
        traffic = None
        pmc_stale = False      # the committed counter files were taken on another build of the library than the one timed here
        try:
            from comfyui_sdmatte_amd import build as _build
            build_stamp = _build._stamp()
        except Exception:
            build_stamp = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                pt = json.load(fh)
            if pt.get("batch_per_gpu") == B and pt.get("inference_size") == S and pt.get("precision", "fp16") == precision:
                traffic = pt.get("conv3x3_bytes_per_launch")
                pmc_stale = pmc_stale or pt.get("build_stamp") != build_stamp
        except Exception:
            pass
        # matrix-pipe busy fraction of the same kernel from the committed SQ counter pass of `bench.py --timed-only`
        # (profiles/pmc_sq.json, written by tools/pmc_sq.py: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)); null when absent
        mfma_busy = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_sq.json")) as fh:
                ps = json.load(fh)
            if ps.get("batch_per_gpu") == B and ps.get("inference_size") == S and ps.get("precision", "fp16") == precision:
                mfma_busy = ps.get("conv3x3_mfma_busy_frac")
                pmc_stale = pmc_stale or ps.get("build_stamp") != build_stamp
        except Exception:
            pass
        f8_res = precision == "fp16x3" and eng.lib.get_option("conv_f8") != 0      # residual terms of the 3x3 convs on fp8 (engine default)
        # matrix-pipe time per algorithmic product in units of one fp16 MFMA: fp16x3 = 3; fp16 + two fp8 residual terms at twice
        # the rate = 2 (a handful of thin / strided launches of the family stay on 3 and are counted as 2: lower bound)
        mfma_per_product = (2 if f8_res else 3) if precision == "fp16x3" else 1
        if "conv3x3_mfma" in prof_exec:
            c = prof_exec["conv3x3_mfma"]
            ach = c["flops"] / (c["ms"] * 1e-3) / 1e12 if c["ms"] > 0 else 0.0
            cd = prof.get("conv3x3_mfma", c)
            ach_d = cd["flops"] / (cd["ms"] * 1e-3) / 1e12 if cd["ms"] > 0 else 0.0
            roof = {"bound": "mfma", "kernel": "conv_mfma_kernel<9,...> (conv3x3 implicit GEMM)", "achieved": round(ach, 2),
                    "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_F16_PEAK_TFLOPS, 4), "traffic": traffic,
                    "launches_per_step": c["launches"], "avg_launch_us": round(c["ms"] * 1e3 / max(c["launches"], 1), 2),
                    "flops_per_launch": c["flops"] / max(c["launches"], 1), "family_ms_per_step": round(c["ms"], 3),
                    "basis": ("EXECUTED flops: this pass runs with trimap_skip = 0, so every tile of every launch is multiplied" if prof_exec is not prof or skip_prev == 0
                              else "dense-equivalent flops (--timed-only keeps every step identical: filled tiles are counted as multiplied)"),
                    "with_constant_tiles_filled": {"achieved_dense_equivalent": round(ach_d, 2), "frac_dense_equivalent": round(ach_d / MFMA_F16_PEAK_TFLOPS, 4),
                                                   "family_ms_per_step": round(cd["ms"], 3),
                                                   "note": "the engine's default (what `value` runs): filled tiles counted as if multiplied - NOT a utilisation figure"},
                    "note": f"achieved = ALGORITHMIC flops (2*MAC of the convolution) / time; this precision keeps the matrix pipe busy for "
                            f"{mfma_per_product} fp16-MFMA time(s) per algorithmic product"
                            + (" (x_hi*w_hi on fp16 + the two residual terms on fp8 K=64 MFMAs at twice the rate)" if f8_res else ""),
                    "mfma_executed_frac": round(mfma_per_product * ach / MFMA_F16_PEAK_TFLOPS, 4), "mfma_busy_frac_pmc": mfma_busy,
                    "pmc_source": "profiles/pmc_traffic.json, profiles/pmc_sq.json (rocprofv3 --pmc passes of `bench.py --timed-only`, reduced by tools/pmc_*.py)",
                    "pmc_stale": pmc_stale, "pmc_measured_in_this_run": False}
            if pmc_stale:      # counters of another build say nothing about this one
                roof["traffic"] = None
                roof["mfma_busy_frac_pmc"] = None
        breakdown = {k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                         "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1) if v["flops"] else None,
                         "gbps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1) if v["bytes"] else None}
                     for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        # ---- the other precision, same inputs, same protocol (N = 1 only): reported, never `value` ----
        modes = {precision: {"images_per_s": round(value, 3), "ms_per_step": round(ms_per_step, 3)}}
        eng_o = None
        if world == 1 and not args.no_other_mode:
            eng_o = E.Engine(cfg, local_rank, precision=other)
            eng_o.load_state_dict(sd)
            step_o = make_step(eng_o)
            for _ in range(max(1, args.warmup)):
                step_o()
            el_o = timed_steps(step_o, args.steps, 1, dev)
            modes[other] = {"images_per_s": round(B * args.steps / el_o, 3), "ms_per_step": round(el_o * 1e3 / args.steps, 3)}
        # ---- CPU baseline + parity: the fp32 oracle on a bounded sample (rank 0, N = 1 only) ----
        cpu = None
        parity = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import sdmatte_oracle as O
            Sc = args.cpu_sample_size
            # the oracle's input IS image 0 of the timed batch (when the sample has the benchmark's own size): `parity` then describes the
            # kernels the headline number ran - tile and kernel selection depend on the batch size
            from_batch = Sc == S
            ci, ct = (img[:1].contiguous(), tri[:1].contiguous()) if from_batch else synthetic_inputs(1, Sc, Sc, seed=1234)
            data = O.preprocess(ci, ct, Sc, False)
            tc = time.perf_counter()
            ref = O.sdmatte_forward(sd, cfg.as_dict(), data)
            tcpu = time.perf_counter() - tc
            scale = FLOPS_PER_IMAGE.get(S, 28.89e12) / FLOPS_PER_IMAGE.get(Sc, 5.96e12)
            cpu = {"value": round(1.0 / (tcpu * scale), 5), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                   "extrapolated": Sc != S,
                   "sample": f"1 image at {Sc}x{Sc} through oracle/sdmatte_oracle.py (fp32 torch CPU restatement of the reference "
                             f"force_cpu path) took {tcpu:.1f} s on this host"
                             + ("" if Sc == S else f"; EXTRAPOLATED to {S}x{S} by the dense-FLOP ratio {scale:.2f}")}
            parity = {"tolerance": 1e-3, "sample": f"{Sc}x{Sc}, full architecture, same synthetic weights, vs the fp32 oracle"}
            if from_batch:
                step()                                   # (the dense-attention / other-mode legs above may have reused the buffer)
                torch.cuda.synchronize()
                dt0 = (alpha[0].detach().float().cpu() - ref[0].clamp(0.0, 1.0).reshape(alpha[0].shape)).abs()
                parity["sample"] = (f"image 0 of the timed batch (B = {B} at {S}x{S}: the kernels the headline number ran), full architecture, same "
                                    "synthetic weights, vs the fp32 oracle clamped to [0, 1] as the node does")
                parity["timed_batch_image0_max_abs_dalpha"] = float(dt0.max())
                parity["timed_batch_image0_mean_abs_dalpha"] = float(dt0.mean())
            for name, en in ((precision, eng), (other, eng_o)):
                if en is None:
                    continue
                got = en.forward(data["image"].to(dev), data["trimap"].to(dev)).cpu()
                dd = (got - ref).abs()
                modes[name]["max_abs_dalpha"] = float(dd.max())
                modes[name]["mean_abs_dalpha"] = float(dd.mean())
            parity["max_abs_dalpha"] = parity.get("timed_batch_image0_max_abs_dalpha", modes[precision]["max_abs_dalpha"])
            parity["single_image_call_max_abs_dalpha"] = modes[precision]["max_abs_dalpha"]
            parity["mean_abs_dalpha"] = modes[precision]["mean_abs_dalpha"]
            parity["within_tolerance"] = parity["max_abs_dalpha"] <= 1e-3
            cpu["max_abs_dalpha_vs_gpu"] = parity["max_abs_dalpha"]
        if eng_o is not None:
            eng_o.close()
        result = {
            "metric": "alpha mattes/sec, mixed-resolution stream (512/768/1024)" if args.stream else "alpha mattes/sec at 1024x1024",
            "value": round(value, 3), "unit": "images/s", "n_gpus": world,
            "ranks_seen": len(per_rank_s),
            "per_rank_images_per_s": None if args.stream else [round(B * args.steps / max(x, 1e-9), 3) for x in per_rank_s],
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16x3" if precision == "fp16x3" else "f16", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[4]: mixed-resolution request stream, inference sizes {sorted(set(stream_sizes))} in turn, "
                                    f"{args.stream_requests} requests per GPU per step ({len(stream_sizes)} in all), FLOP-balanced buckets per rank, micro-batches "
                                    f"of <= {B} equal-size images, matted_rgba output (node body incl. composition on the GPU), one packed message per peer to rank 0; "
                                    "SD-2.1/SDMatte architecture, synthetic weights; kernel_breakdown_ms / roofline below describe ONE uniform 1024^2 batch, not the stream")
                       if args.stream else
                       f"BASELINE configs[1]: {S}x{S} image+trimap -> alpha (alpha_only), SD-2.1/SDMatte architecture, "
                       f"synthetic weights, {B} images per GPU per step",
                       "inference_size": S, "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "precision": precision,
                       "arithmetic": (("split MFMA operands x = hi + lo, fp32 accumulate, fp32 activations: x_hi*w_hi on fp16; the residual "
                                       "terms x_lo*w + x*w_lo on fp8 (one K=64 MFMA; e5m2 activations x e4m3 weights with a per-layer scale) in the "
                                       "3x3 convs with >= 128 output channels, the GEMMs with K >= 1024 and (e5m2 x e5m2) Q.K^T of the d=64 "
                                       "attention cores, on fp16 (2 more MFMAs) in every other conv / GEMM; P.V and the softmax denominator on the "
                                       "same plain fp16 probabilities (fp32 accumulate); the d=512 VAE attention core on plain fp16"
                                       if eng.lib.get_option("conv_f8") != 0 else
                                       "split-fp16 MFMA operands (hi+lo, 3 MFMAs per product), fp32 accumulate, fp32 activations")
                                      if precision == "fp16x3" else "fp16 MFMA operands, fp32 accumulate, fp32 residual stream"),
                       "trimap": "synthetic disc/annulus (28 % foreground / 22 % unknown / 50 % background, SURVEY.md 8d)",
                       "trimap_encoder_tiles": ("output tiles of the VAE encoder's wide 3x3 convs that lie inside a constant region of the trimap image are filled from "
                                                "one multiplied tile per (image, region value) - exact; engine option trimap_skip = 0 multiplies all")
                       if eng.lib.get_option("trimap_skip") != 0 and precision == "fp16x3" else "every tile multiplied",
                       "self_attention_keys": "all key tiles" if args.dense_attention else
                       "key tiles whose (1-m)*-10000 bias underflows the fp32 softmax are not loaded (exact; --dense-attention disables)"},
            "parity": parity, "modes": modes, "dense_attention": dense, "every_trimap_tile_multiplied": all_tiles, "no_input_shortcuts": no_short,
            "launches_per_step": sum(v["launches"] for v in prof.values()), "single_image": b1, "including_host_transfers": incl,
            # dense-equivalent algorithmic rate (28.89 TFLOP per 1024^2 image); the self-attention skips the key tiles
            # whose bias underflows the softmax, so the executed attention work depends on the trimap (kernel_breakdown_ms has
            # executed rates per kernel)
            "tflops_per_gpu": round((sum(FLOPS_PER_IMAGE.get(x, 0) for x in stream_sizes) / world if args.stream else FLOPS_PER_IMAGE.get(S, 0) * B)
                                    / (ms_per_step * 1e-3) / 1e12, 1),
            "tflops_per_gpu_basis": "dense-equivalent algorithmic FLOPs of the reference graph (not executed FLOPs)",
            "weight_load_s": round(load_s, 1),
            "roofline": roof, "cpu_baseline": cpu, "kernel_breakdown_ms": breakdown,
        }
        print(json.dumps(result))
        sys.stdout.flush()
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
