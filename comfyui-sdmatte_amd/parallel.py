"""Image-level data parallelism for the SDMatte path: one process per GPU under torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm).  Images are independent (no cross-image reduction anywhere in the graph), so a
batch is split contiguously across ranks and the only collectives are
  * broadcast of the packed fp16 weight blob (+ the small host-side embedding tensors) from the rank that read the
    checkpoint - once per checkpoint, and
  * gather of the per-rank alphas to the rank that returns them.
The reference has no distributed code at all; this is new, not a translation."""
import torch
import torch.distributed as dist


def shard_range(total: int, world: int, rank: int):
    """Contiguous split: rank r gets images [r*ceil(total/world), ...)."""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def broadcast_weights(engine, src: int = 0, device=None):
    """Rank `src` has loaded a checkpoint into `engine`; every other rank receives the packed weights: ONE RCCL broadcast of
    [packed device arena | small host-side embedding tensors] (the host part rides at the tail of the same buffer)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    rank = dist.get_rank()
    device = device if device is not None else torch.device("cuda", engine.device)
    nw, nh = engine.weight_blob_bytes(), engine.host_blob_bytes()
    buf = torch.empty(nw + nh, dtype=torch.uint8, device=device)
    if rank == src:
        hblob = torch.empty(nh, dtype=torch.uint8)
        engine.export_weights(buf[:nw], hblob)
        buf[nw:].copy_(hblob)
    dist.broadcast(buf, src)
    if rank != src:
        hblob = buf[nw:].cpu()
        if buf.is_cuda:
            torch.cuda.current_stream(buf.device).synchronize()      # the import reads `buf` on the engine's own stream: the broadcast must have landed
        engine.import_weights(buf[:nw], hblob)


def gather_alphas(alpha: torch.Tensor, dst: int = 0):
    """Equal-shape gather (uniform batches): returns the list of per-rank alphas on `dst`, None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [alpha]
    out = [torch.empty_like(alpha) for _ in range(dist.get_world_size())] if dist.get_rank() == dst else None
    dist.gather(alpha, out, dst=dst)
    return out


FLOPS_PER_IMAGE = {512: 5.96, 640: 9.6, 768: 14.59, 896: 21.0, 1024: 28.89}   # TFLOP, SURVEY.md 8d (640/896 interpolated ~S^2.3)


def bucket_requests(sizes, world: int):
    """Mixed-resolution stream (BASELINE config #5): bucket request indices by inference_size and assign whole buckets'
    images to ranks greedily by estimated FLOPs (longest-processing-time first).  Returns per-rank {size: [request idx]}."""
    loads = [0.0] * world
    plan = [dict() for _ in range(world)]
    order = sorted(range(len(sizes)), key=lambda i: -FLOPS_PER_IMAGE.get(int(sizes[i]), (int(sizes[i]) / 1024.0) ** 2.3 * 28.89))
    for i in order:
        r = min(range(world), key=lambda k: loads[k])
        s = int(sizes[i])
        plan[r].setdefault(s, []).append(i)
        loads[r] += FLOPS_PER_IMAGE.get(s, (s / 1024.0) ** 2.3 * 28.89)
    for p in plan:
        for s in p:
            p[s].sort()
    return plan


def matte_stream(engine, images, trimaps, sizes, is_transparent: bool = False, micro_batch: int = 4, dst: int = 0, device=None,
                 output_mode=None, mask_refine: bool = False, trimap_constraint: float = 0.8):
    """Mixed-resolution request stream (BASELINE config #5): request i = (images[i] [H,W,3], trimaps[i] [H,W], sizes[i]).
    Every rank holds the request list; `bucket_requests` gives each rank whole same-size groups balanced by estimated FLOPs;
    a rank runs its groups in micro-batches of equal (H, W) and the alphas travel to `dst` point to point, one packed message per
    peer (shapes are known from the request list, so no size exchange and no padding).  Returns the list of alphas [H,W] in request order on `dst`,
    None on the other ranks.  With `output_mode` "matted_rgba" / "matted_rgb" (BASELINE config #5 asks for matted_rgba) every request goes through the
    whole node body (`apply_matte_node`: mask_refine and the output composition on the GPU too) and what travels and is returned is the composed image
    [H,W,4|3] (its last channel is the alpha for matted_rgba).  No collective besides these sends exists on the path."""
    chans = None
    if output_mode is not None and output_mode != "alpha_only":
        chans = {"matted_rgba": 4, "matted_rgb": 3}[output_mode]
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    device = device if device is not None else torch.device("cuda", engine.device)
    plan = bucket_requests(sizes, world)
    mine = {}
    for S, idxs in sorted(plan[rank].items()):
        by_shape = {}
        for i in idxs:
            by_shape.setdefault(tuple(images[i].shape[:2]), []).append(i)
        for _, ids in sorted(by_shape.items()):
            for k in range(0, len(ids), micro_batch):
                chunk = ids[k:k + micro_batch]
                img = torch.stack([images[i] for i in chunk]).to(device)
                tri = torch.stack([trimaps[i] for i in chunk]).to(device)
                if chans is None:
                    a = engine.apply_matte(img, tri, S, is_transparent)
                else:
                    a = engine.apply_matte_node(img, tri, S, is_transparent, output_mode, mask_refine, trimap_constraint)[1]
                for j, i in enumerate(chunk):
                    mine[i] = a[j]
    if world == 1:
        return [mine[i] for i in range(len(sizes))]
    order = lambda r: sorted(i for v in plan[r].values() for i in v)      # the same deterministic order on both ends
    shape = lambda i: (int(images[i].shape[0]), int(images[i].shape[1])) + (() if chans is None else (chans,))
    numel = lambda i: int(images[i].shape[0]) * int(images[i].shape[1]) * (chans or 1)
    # ONE message per peer: a rank's alphas travel as one flat fp32 buffer (shapes are known on both ends from the request list), and `dst` posts the
    # receives of all peers as ONE batch (dist.batch_isend_irecv: a single grouped launch on the NCCL / RCCL backend - un-batched point-to-point
    # operations there may serialise per peer communicator), so the seven peers of an 8-GPU node drain concurrently over their own xGMI links
    if rank != dst:
        ids = order(rank)
        if ids:
            flat = torch.cat([mine[i].reshape(-1).to(torch.float32) for i in ids]) if len(ids) > 1 else mine[ids[0]].reshape(-1).to(torch.float32).contiguous()
            for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, flat, dst)]):
                q.wait()
        return None
    out = [None] * len(sizes)
    for i, a in mine.items():
        out[i] = a
    ops, bufs = [], {}
    for r in range(world):
        ids = order(r) if r != dst else []
        if not ids:
            continue
        bufs[r] = torch.empty(sum(numel(i) for i in ids), dtype=torch.float32, device=device)
        ops.append(dist.P2POp(dist.irecv, bufs[r], r))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    for r, flat in bufs.items():
        off = 0
        for i in order(r):
            n = numel(i)
            out[i] = flat[off:off + n].view(*shape(i))
            off += n
    return out


class MultiGpuEngine:
    """Single-process fan-out over the GPUs of one node (ComfyUI is ONE process, its node is called from one
    worker thread - reference sdmatte_nodes.py:257): one engine, one HIP stream and one host thread per device.  The checkpoint
    is packed once on the first device and the packed arena is copied device to device (xGMI peer copies); a batch is split
    contiguously (`shard_range`), every shard goes through its device's engine concurrently (the C ABI releases the GIL), and
    the alphas land in one host tensor.  No collective exists on this path: images are independent."""

    def __init__(self, cfg=None, devices=None, precision=None, stream_f32: bool = True, _engine_factory=None):
        from .engine import Engine
        if devices is None:
            devices = list(range(torch.cuda.device_count()))
        if not devices:
            raise RuntimeError("MultiGpuEngine: no GPU given")
        make = _engine_factory or (lambda d: Engine(cfg, d, stream_f32, precision=precision))
        self.devices = list(devices)
        self.engines = [make(d) for d in self.devices]
        self._owned = list(self.engines)
        self.device = self.devices[0]
        self.cfg = self.engines[0].cfg
        self._on_device = self.engines[0]._on_device

    @classmethod
    def around(cls, first, other_devices):
        """Fan-out around an engine that already holds a checkpoint (the node's cached model): engines for `other_devices` are
        created with the same configuration / precision and receive the packed weights device to device."""
        from .engine import Engine
        self = cls.__new__(cls)
        self.devices = [first.device] + list(other_devices)
        self.engines = [first] + [Engine(first.cfg, d, bool(first._ccfg.stream_f32), precision=first.precise_mask) for d in other_devices]
        self._owned = self.engines[1:]
        self.device, self.cfg, self._on_device = first.device, first.cfg, first._on_device
        self._copy_weights()
        return self

    def close(self):
        for e in self._owned:
            e.close()
        self.engines, self._owned = [], []

    def load_state_dict(self, state_dict, strict: bool = False):
        res = self.engines[0].load_state_dict(state_dict, strict)
        self._copy_weights()
        return res

    def _copy_weights(self):
        """Packed (canonical) weight arena of engine 0 -> every other engine.  The peer copies run concurrently, one torch stream
        per destination (xGMI is point to point: every peer has its own link to device 0), and every copy has COMPLETED before
        the receiving engine imports the blob: `sdm_import_weight_blob` reads it on the engine's own non-blocking stream, which
        is not ordered against torch's streams."""
        e0 = self.engines[0]
        if len(self.engines) < 2:
            return
        hblob = torch.empty(e0.host_blob_bytes(), dtype=torch.uint8)
        if not self._on_device:                                    # emulator engines (tests): host memory
            blob = torch.empty(e0.weight_blob_bytes(), dtype=torch.uint8)
            e0.export_weights(blob, hblob)
            for e in self.engines[1:]:
                e.import_weights(blob, hblob)
            return
        dev0 = torch.device("cuda", self.devices[0])
        blob = torch.empty(e0.weight_blob_bytes(), dtype=torch.uint8, device=dev0)
        e0.export_weights(blob, hblob)                             # returns after a host sync of engine 0's stream
        peers, streams = [], []
        for d in self.devices[1:]:
            st = torch.cuda.Stream(device=dev0)
            with torch.cuda.stream(st):
                peers.append(blob.to(torch.device("cuda", d), non_blocking=True))      # hipMemcpyPeerAsync over that peer's xGMI link
            streams.append(st)
        for st, d in zip(streams, self.devices[1:]):
            st.synchronize()
            torch.cuda.synchronize(d)
        for e, peer in zip(self.engines[1:], peers):
            e.import_weights(peer, hblob)

    def _host(self, *shape):
        """Host tensor the engines copy into / out of: page-locked when GPUs are driven (pageable memory would make every shard's
        hipMemcpyAsync a staged, blocking copy), plain otherwise."""
        return torch.empty(*shape, dtype=torch.float32, pin_memory=self._on_device)

    def _to_host(self, t):
        t = t.detach().float()
        if t.device.type == "cpu" and (not self._on_device or t.is_pinned()):
            return t.contiguous()
        buf = self._host(*t.shape)
        buf.copy_(t)
        return buf

    def _fan(self, B, call):
        import threading
        n = min(len(self.engines), B)
        errs = [None] * n

        def work(r):
            try:
                lo, hi = shard_range(B, n, r)
                if hi > lo:
                    call(self.engines[r], self.devices[r], lo, hi)
            except BaseException as exc:  # noqa: BLE001 - re-raised on the calling thread
                errs[r] = exc

        threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(1, n)]
        for t in threads:
            t.start()
        work(0)
        for t in threads:
            t.join()
        for exc in errs:
            if exc is not None:
                raise exc

    def apply_matte(self, image_bhwc, trimap_bhw, S, is_transparent=False):
        """image [B,H,W,3], trimap [B,H,W] (host or any device) -> alpha [B,H,W] fp32 on the HOST (what the node returns)."""
        B, H, W, _ = image_bhwc.shape
        out = self._host(B, H, W)
        img = self._to_host(image_bhwc)
        tri = self._to_host(trimap_bhw)

        def call(eng, dev, lo, hi):
            # host pointers: the engine copies its shard in on its own stream, runs, copies the alphas out and synchronises
            eng.apply_matte(img[lo:hi], tri[lo:hi], S, is_transparent, out=out[lo:hi])

        self._fan(B, call)
        return out

    def apply_matte_node(self, image_bhwc, trimap_bhw, S, is_transparent, output_mode, mask_refine, trimap_constraint):
        """The whole node body (incl. mask_refine + output composition on each GPU) for a batch split over the devices;
        returns (alpha [B,H,W], matted [B,H,W,3|4]) on the HOST."""
        from .engine import Engine
        B, H, W, _ = image_bhwc.shape
        ch = 4 if Engine.OUTPUT_MODES[output_mode] == 1 else 3
        alpha = self._host(B, H, W)
        matted = self._host(B, H, W, ch)
        img = self._to_host(image_bhwc)
        tri = self._to_host(trimap_bhw)

        def call(eng, dev, lo, hi):
            a, m = eng.apply_matte_node(img[lo:hi], tri[lo:hi], S, is_transparent, output_mode, mask_refine, trimap_constraint)
            alpha[lo:hi].copy_(a)
            matted[lo:hi].copy_(m)

        self._fan(B, call)
        return alpha, matted

    def last_forward_ms(self):
        return max(e.last_forward_ms() for e in self.engines)
