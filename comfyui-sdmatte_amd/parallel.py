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
    """Rank `src` has loaded a checkpoint into `engine`; every other rank receives the packed blob."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    rank = dist.get_rank()
    device = device if device is not None else torch.device("cuda", engine.device)
    blob = torch.empty(engine.weight_blob_bytes(), dtype=torch.uint8, device=device)
    hblob = torch.empty(engine.host_blob_bytes(), dtype=torch.uint8)
    if rank == src:
        engine.export_weights(blob, hblob)
    dist.broadcast(blob, src)
    hb = hblob.to(device)
    dist.broadcast(hb, src)
    if rank != src:
        engine.import_weights(blob, hb.cpu())


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


def matte_stream(engine, images, trimaps, sizes, is_transparent: bool = False, micro_batch: int = 4, dst: int = 0, device=None):
    """Mixed-resolution request stream (BASELINE config #5): request i = (images[i] [H,W,3], trimaps[i] [H,W], sizes[i]).
    Every rank holds the request list; `bucket_requests` gives each rank whole same-size groups balanced by estimated FLOPs;
    a rank runs its groups in micro-batches of equal (H, W) and the alphas travel to `dst` point to point (shapes are known
    from the request list, so no size exchange and no padding).  Returns the list of alphas [H,W] in request order on `dst`,
    None on the other ranks.  No collective besides these sends exists on the path."""
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
                a = engine.apply_matte(img, tri, S, is_transparent)
                for j, i in enumerate(chunk):
                    mine[i] = a[j]
    if world == 1:
        return [mine[i] for i in range(len(sizes))]
    order = lambda r: sorted(i for v in plan[r].values() for i in v)      # the same deterministic order on both ends
    if rank != dst:
        for i in order(rank):
            dist.send(mine[i].contiguous(), dst)
        return None
    out = [None] * len(sizes)
    for i, a in mine.items():
        out[i] = a
    for r in range(world):
        if r == dst:
            continue
        for i in order(r):
            buf = torch.empty(tuple(images[i].shape[:2]), dtype=torch.float32, device=device)
            dist.recv(buf, r)
            out[i] = buf
    return out
