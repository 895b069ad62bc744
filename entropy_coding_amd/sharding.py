"""Substream sharding for one-process-per-GPU runs.

CABAC substreams (slices / tiles / frames) are independent units: private context store, private
low/range, private byte stream (reference cabac_writer.cpp:16-39, :104-107); the only cross-unit
operation is the ordered concatenation of the finished byte strings (bit_stream.cpp:139-150).  So the
data path has NO collective: each rank codes its own substreams.  torch.distributed (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in CPU tests) moves the bin records from the ingest rank to the
ranks and the coded bytes back:

* payloads stay where the backend wants them — device tensors under RCCL (records are cut out of the
  root's device buffer and land in the receivers' device buffers; the coded bytes travel back device to
  device), CPU tensors under gloo;
* the sizes of everything a transfer will carry are exchanged ONCE, as a collective (broadcast of the
  root's table for the scatter, all_gather for the gather), not as a message in front of every array;
* the point-to-point transfers of one scatter (or gather) are posted TOGETHER with
  dist.batch_isend_irecv and waited for once: on a fully connected 8-GPU xGMI node the root's seven
  links then carry their transfers at the same time (a ring collective would be bound by one link,
  blocking sends one after another by the sum).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import capi


def lpt_assign(n_records, world):
    """Longest-processing-time-first bin packing of substreams onto `world` ranks by bin count.
    Returns a list (per rank) of substream indices, each LONGEST FIRST (equal lengths in their original order): a wave holds
    consecutive substreams of a shard and runs as long as its longest, and the decode dispatch gives a shard whose few long
    substreams come first a wave per substream (csrc/cabac_kernels_v4.hip, decode_select_solo_kernel)."""
    order = np.argsort(-np.asarray(n_records, np.int64), kind="stable")
    load = np.zeros(world, np.int64)
    owner = [[] for _ in range(world)]
    for s in order:
        r = int(np.argmin(load))
        owner[r].append(int(s))
        load[r] += int(n_records[s])
    return owner


def _dev():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _as_bytes(a):
    """numpy array or torch tensor -> flat uint8 tensor on the transport device (a view where it already lives there)."""
    if isinstance(a, torch.Tensor):
        return a.contiguous().view(-1).view(torch.uint8).to(_dev())
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(_dev())


def _to_numpy(t, dtype):
    return t.cpu().numpy().view(dtype)


# ---- transport ---------------------------------------------------------------------------------------------------
def star_scatter(parts, n_parts, root=0):
    """Root holds parts[r] = [tensor-like, ...] (n_parts of them) for every rank r; every rank gets its own list of flat
    uint8 tensors on the transport device.  One broadcast of the size table, then all transfers in one batch."""
    rank, world = dist.get_rank(), dist.get_world_size()
    table = torch.zeros(world * n_parts, dtype=torch.int64, device=_dev())
    if rank == root:
        parts = [[_as_bytes(p) for p in parts[r]] for r in range(world)]
        table = torch.tensor([p.numel() for r in range(world) for p in parts[r]], dtype=torch.int64, device=_dev())
    dist.broadcast(table, root)
    sizes = table.cpu().numpy().reshape(world, n_parts)
    ops, mine = [], None
    if rank == root:
        mine = parts[root]
        for r in range(world):
            if r != root:
                ops += [dist.P2POp(dist.isend, p, r) for p in parts[r] if p.numel()]
    else:
        mine = [torch.empty(int(n), dtype=torch.uint8, device=_dev()) for n in sizes[rank]]
        ops = [dist.P2POp(dist.irecv, t, root) for t in mine if t.numel()]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return mine


def star_gather(my_parts, root=0):
    """Every rank hands in a list of tensor-likes; root gets [[flat uint8 tensors of rank 0], [of rank 1], ...].
    One all_gather of the sizes, then all transfers in one batch."""
    rank, world = dist.get_rank(), dist.get_world_size()
    my_parts = [_as_bytes(p) for p in my_parts]
    n_parts = len(my_parts)
    mine = torch.tensor([p.numel() for p in my_parts], dtype=torch.int64, device=_dev())
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine)
    ops, got = [], None
    if rank == root:
        got = []
        for r in range(world):
            if r == root:
                got.append(my_parts)
                continue
            bufs = [torch.empty(int(n), dtype=torch.uint8, device=_dev()) for n in sizes[r].cpu().numpy()]
            ops += [dist.P2POp(dist.irecv, t, r) for t in bufs if t.numel()]
            got.append(bufs)
    else:
        ops = [dist.P2POp(dist.isend, p, root) for p in my_parts if p.numel()]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return got


# ---- substreams --------------------------------------------------------------------------------------------------
def pack_shard(desc, records, idxs):
    """Re-pack the substreams `idxs` of (desc, records) into a self-contained shard: (desc, records, bytes_total).
    `records` may be a numpy array or a torch tensor (e.g. the root's device buffer): the shard's records are of the
    same kind."""
    sub = desc[idxs].copy()
    lens = sub["n_records"].astype(np.int64)
    caps = sub["byte_capacity"].astype(np.int64)
    sub["rec_offset"] = np.concatenate([[0], np.cumsum(lens)[:-1]]) if len(idxs) else []
    sub["byte_offset"] = np.concatenate([[0], np.cumsum((caps + 15) // 16 * 16)[:-1]]) if len(idxs) else []
    if isinstance(records, torch.Tensor) and records.is_cuda and len(idxs):
        # one gather launch on the device (cabac_hip_gather_records_device) instead of one slice per substream and a cat
        total = int(lens.sum())
        recs = torch.empty(total, dtype=records.dtype, device=records.device)
        dev = records.device
        src = torch.from_numpy(desc["rec_offset"][idxs].astype(np.int64)).to(dev)
        dst = torch.from_numpy(sub["rec_offset"].astype(np.int64)).to(dev)
        ln = torch.from_numpy(lens.astype(np.int32)).to(dev)
        hip = _gather_ctx(dev.index if dev.index is not None else torch.cuda.current_device())
        hip.gather_records_device(len(idxs), src.data_ptr(), dst.data_ptr(), ln.data_ptr(), records.data_ptr(), recs.data_ptr())
        hip.synchronize()   # src / dst / ln are this function's: they must outlive the launch whatever stream it ran on
        return sub, recs, int(((caps + 15) // 16 * 16).sum())
    pieces = [records[int(desc["rec_offset"][i]):int(desc["rec_offset"][i]) + int(desc["n_records"][i])] for i in idxs]
    if isinstance(records, torch.Tensor):
        recs = torch.cat(pieces) if pieces else records[:0]
    else:
        recs = np.concatenate(pieces) if pieces else np.zeros(0, np.uint16)
    return sub, recs, int(((caps + 15) // 16 * 16).sum())


_gather = {}


def _gather_ctx(device):
    """A codec context on torch's current stream of `device` for the gather launches (kept: one per device)."""
    stream = torch.cuda.current_stream(device).cuda_stream
    key = (device, stream)
    if key not in _gather:
        _gather[key] = capi.CabacHip(device, stream=stream)
    return _gather[key]


def scatter_substreams(desc, records, root=0):
    """Root holds the whole batch (desc: numpy; records: numpy or a torch tensor); every rank returns its shard
    (desc numpy, records flat uint8 tensor on the transport device — 2 bytes per record —, bytes_total, global indices)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    parts = None
    if rank == root:
        owners = lpt_assign(desc["n_records"], world)
        parts = []
        for r in range(world):
            d, rec, _ = pack_shard(desc, records, owners[r])
            parts.append([d, rec, np.asarray(owners[r], np.int64)])
    d, rec, idx = star_scatter(parts, 3, root)
    d = _to_numpy(d, capi.DESC_DTYPE).copy()
    total = int(((d["byte_capacity"].astype(np.int64) + 15) // 16 * 16).sum())
    return d, rec, total, _to_numpy(idx, np.int64).copy()


def gather_payloads(idx, results, payload, root=0):
    """Every rank hands in the global indices of its substreams, their results and their coded bytes back to back
    (numpy or tensor).  Root returns [(idx, results, payload tensor)] per rank; the others None."""
    got = star_gather([np.asarray(idx, np.int64), results, payload], root)
    if got is None:
        return None
    return [(_to_numpy(g[0], np.int64), _to_numpy(g[1], capi.RESULT_DTYPE), g[2]) for g in got]


def ordered_streams(n_total, gathered):
    """(list of per-substream byte arrays in global order, n_bits array) from gather_payloads' result."""
    streams = [None] * n_total
    n_bits = np.zeros(n_total, np.uint32)
    for ridx, rres, rpay in gathered:
        pay = rpay.cpu().numpy()
        rn = (rres["n_bits"].astype(np.int64) + 7) // 8
        off = np.concatenate([[0], np.cumsum(rn)])
        for k, g in enumerate(ridx):
            streams[int(g)] = pay[int(off[k]):int(off[k + 1])].copy()
            n_bits[int(g)] = rres["n_bits"][k]
    return streams, n_bits


def compact_payload(desc, out_bytes, results):
    """The coded substreams of a shard back to back (host arrays; on the device cabac_hip_assemble_device does this)."""
    nbytes = (results["n_bits"].astype(np.int64) + 7) // 8
    return np.concatenate([out_bytes[int(desc["byte_offset"][k]):int(desc["byte_offset"][k]) + int(nbytes[k])]
                           for k in range(len(desc))]) if len(desc) else np.zeros(0, np.uint8)


def encode_sharded(desc, records, encode_fn, root=0):
    """Scatter -> encode locally with encode_fn(desc, records, bytes_total) -> (bytes, results) -> gather.
    `encode_fn` is CabacHip.encode_batch on a GPU box; tests inject a CPU checker.  Returns on root
    (list of per-substream byte arrays, n_bits array); elsewhere None."""
    n_total = torch.tensor([len(desc) if dist.get_rank() == root else 0], dtype=torch.int64, device=_dev())
    dist.broadcast(n_total, root)
    d, rec, total, idx = scatter_substreams(desc, records, root)
    out, res = encode_fn(d, _to_numpy(rec, np.uint16), total)
    got = gather_payloads(idx, res, compact_payload(d, out, res), root)
    return None if got is None else ordered_streams(int(n_total.item()), got)


# ---- residual binariser: transform blocks shard the same way ------------------------------------
# The unit is a tile (a contiguous group of blocks that ends up in one substream); blocks are independent of each other,
# so any grouping is correct — tiles keep a substream's records on one rank for the encoder that follows.
def pack_tiles(tus, coeff, tile_first, tiles):
    """Re-pack the blocks of `tiles` into a self-contained shard: (tus, coeff, global block indices)."""
    blk = np.concatenate([np.arange(int(tile_first[t]), int(tile_first[t + 1]), dtype=np.int64) for t in tiles]) \
        if len(tiles) else np.zeros(0, np.int64)
    sub = tus[blk].copy()
    sizes = (1 << (sub["log2_width"].astype(np.int64) + sub["log2_height"].astype(np.int64)))
    new_off = np.concatenate([[0], np.cumsum(sizes)[:-1]]) if len(blk) else np.zeros(0, np.int64)
    co = np.concatenate([coeff[int(tus["coeff_offset"][b]): int(tus["coeff_offset"][b]) + int(n)] for b, n in zip(blk, sizes)]) \
        if len(blk) else np.zeros(0, np.int32)
    sub["coeff_offset"] = new_off
    return sub, co.astype(np.int32), blk


def residual_sharded(tus, coeff, tile_first, residual_fn, root=0):
    """Scatter tiles (LPT by coefficient count) -> residual_fn(tus, coeff) -> (records, offsets, info) on every rank ->
    gather on root in global block order.  Returns on root (list of per-block record arrays, info array); else None.
    `residual_fn` is CabacHip.residual_batch on a GPU box; tests inject a CPU checker."""
    rank, world = dist.get_rank(), dist.get_world_size()
    n_total = torch.tensor([len(tus) if rank == root else 0], dtype=torch.int64, device=_dev())
    dist.broadcast(n_total, root)
    parts = None
    if rank == root:
        sizes = (1 << (tus["log2_width"].astype(np.int64) + tus["log2_height"].astype(np.int64)))
        csum = np.concatenate([[0], np.cumsum(sizes)])
        weight = csum[np.asarray(tile_first[1:], np.int64)] - csum[np.asarray(tile_first[:-1], np.int64)]
        owners = lpt_assign(weight, world)
        parts = [list(pack_tiles(tus, coeff, tile_first, owners[r])) for r in range(world)]
    sub, co, blk = star_scatter(parts, 3, root)
    sub, co, blk = _to_numpy(sub, capi.TU_DTYPE).copy(), _to_numpy(co, np.int32).copy(), _to_numpy(blk, np.int64).copy()
    rec, off, info = residual_fn(sub, co) if len(sub) else (np.zeros(0, np.uint16), np.zeros(1, np.uint64), np.zeros(0, np.uint32))
    got = star_gather([blk, np.asarray(off, np.uint64), np.asarray(info, np.uint32), np.asarray(rec, np.uint16)], root)
    if got is None:
        return None
    out = [None] * int(n_total.item())
    infos = np.zeros(int(n_total.item()), np.uint32)
    for g in got:
        rblk, roff, rinfo, rrec = _to_numpy(g[0], np.int64), _to_numpy(g[1], np.uint64), _to_numpy(g[2], np.uint32), _to_numpy(g[3], np.uint16)
        for k, b in enumerate(rblk):
            out[int(b)] = rrec[int(roff[k]): int(roff[k + 1])].copy()
            infos[int(b)] = rinfo[k]
    return out, infos
