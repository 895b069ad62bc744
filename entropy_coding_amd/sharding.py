"""Substream sharding for one-process-per-GPU runs.

CABAC substreams (slices / tiles / frames) are independent units: private context store, private
low/range, private byte stream (reference cabac_writer.cpp:16-39, :104-107); the only cross-unit
operation is the ordered concatenation of the finished byte strings (bit_stream.cpp:139-150).  So the
data path has NO collective: each rank codes its own substreams.  torch.distributed (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in CPU tests) is used only for the scatter of the bin records from
the ingest rank and the gather of the variable-length results, as point-to-point send/recv pairs: on a
fully connected 8-GPU xGMI node a direct star uses all 7 links of the root concurrently, whereas a ring
collective would be bound by one link.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import capi


def lpt_assign(n_records, world):
    """Longest-processing-time-first bin packing of substreams onto `world` ranks by bin count.
    Returns a list (per rank) of substream indices, each in ascending order."""
    order = np.argsort(-np.asarray(n_records, np.int64), kind="stable")
    load = np.zeros(world, np.int64)
    owner = [[] for _ in range(world)]
    for s in order:
        r = int(np.argmin(load))
        owner[r].append(int(s))
        load[r] += int(n_records[s])
    return [sorted(o) for o in owner]


def _dev():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _send_array(a, dst):
    t = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(_dev())
    n = torch.tensor([t.numel()], dtype=torch.int64, device=_dev())
    dist.send(n, dst)
    if t.numel():
        dist.send(t, dst)


def _recv_array(src, dtype):
    n = torch.zeros(1, dtype=torch.int64, device=_dev())
    dist.recv(n, src)
    t = torch.empty(int(n.item()), dtype=torch.uint8, device=_dev())
    if t.numel():
        dist.recv(t, src)
    return t.cpu().numpy().view(dtype)


def pack_shard(desc, records, idxs):
    """Re-pack the substreams `idxs` of (desc, records) into a self-contained shard."""
    sub = desc[idxs].copy()
    lens = sub["n_records"].astype(np.int64)
    caps = sub["byte_capacity"].astype(np.int64)
    sub["rec_offset"] = np.concatenate([[0], np.cumsum(lens)[:-1]]) if len(idxs) else []
    sub["byte_offset"] = np.concatenate([[0], np.cumsum((caps + 15) // 16 * 16)[:-1]]) if len(idxs) else []
    recs = np.concatenate([records[int(desc["rec_offset"][i]):int(desc["rec_offset"][i]) + int(desc["n_records"][i])]
                           for i in idxs]) if len(idxs) else np.zeros(0, np.uint16)
    return sub, recs, int(((caps + 15) // 16 * 16).sum())


def scatter_substreams(desc, records, root=0):
    """Root holds the whole batch; every rank returns its shard (desc, records, bytes_total, global_idx)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = None
    if rank == root:
        owners = lpt_assign(desc["n_records"], world)
        for r in range(world):
            d, rec, total = pack_shard(desc, records, owners[r])
            if r == root:
                mine = (d, rec, total, np.asarray(owners[r], np.int64))
            else:
                _send_array(d, r)
                _send_array(rec, r)
                _send_array(np.asarray(owners[r], np.int64), r)
    else:
        d = _recv_array(root, capi.DESC_DTYPE).copy()
        rec = _recv_array(root, np.uint16).copy()
        idx = _recv_array(root, np.int64).copy()
        total = int(((d["byte_capacity"].astype(np.int64) + 15) // 16 * 16).sum())
        mine = (d, rec, total, idx)
    return mine


def gather_bitstreams(n_total, idx, desc, out_bytes, results, root=0):
    """Gather the coded substreams on root in global substream order.
    Returns on root: (list of per-substream byte arrays, n_bits array); elsewhere None."""
    rank, world = dist.get_rank(), dist.get_world_size()
    nbytes = (results["n_bits"].astype(np.int64) + 7) // 8
    payload = np.concatenate([out_bytes[int(desc["byte_offset"][k]):int(desc["byte_offset"][k]) + int(nbytes[k])]
                              for k in range(len(desc))]) if len(desc) else np.zeros(0, np.uint8)
    if rank != root:
        _send_array(idx, root)
        _send_array(results, root)
        _send_array(payload, root)
        return None
    streams = [None] * n_total
    n_bits = np.zeros(n_total, np.uint32)
    for r in range(world):
        if r == root:
            ridx, rres, rpay = idx, results, payload
        else:
            ridx = _recv_array(r, np.int64)
            rres = _recv_array(r, capi.RESULT_DTYPE)
            rpay = _recv_array(r, np.uint8)
        rn = (rres["n_bits"].astype(np.int64) + 7) // 8
        off = np.concatenate([[0], np.cumsum(rn)])
        for k, g in enumerate(ridx):
            streams[int(g)] = rpay[int(off[k]):int(off[k + 1])].copy()
            n_bits[int(g)] = rres["n_bits"][k]
    return streams, n_bits


def encode_sharded(desc, records, encode_fn, root=0):
    """Scatter -> encode locally with encode_fn(desc, records, bytes_total) -> (bytes, results) -> gather.
    `encode_fn` is CabacHip.encode_batch on a GPU box; tests inject a CPU checker."""
    n_total = torch.tensor([len(desc) if dist.get_rank() == root else 0], dtype=torch.int64, device=_dev())
    dist.broadcast(n_total, root)
    d, rec, total, idx = scatter_substreams(desc, records, root)
    out, res = encode_fn(d, rec, total)
    return gather_bitstreams(int(n_total.item()), idx, d, out, res, root)


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
    if rank == root:
        sizes = (1 << (tus["log2_width"].astype(np.int64) + tus["log2_height"].astype(np.int64)))
        csum = np.concatenate([[0], np.cumsum(sizes)])
        weight = csum[np.asarray(tile_first[1:], np.int64)] - csum[np.asarray(tile_first[:-1], np.int64)]
        owners = lpt_assign(weight, world)
        mine = None
        for r in range(world):
            shard = pack_tiles(tus, coeff, tile_first, owners[r])
            if r == root:
                mine = shard
            else:
                for a in shard:
                    _send_array(a, r)
    else:
        mine = (_recv_array(root, capi.TU_DTYPE).copy(), _recv_array(root, np.int32).copy(), _recv_array(root, np.int64).copy())
    sub, co, blk = mine
    rec, off, info = residual_fn(sub, co) if len(sub) else (np.zeros(0, np.uint16), np.zeros(1, np.uint64), np.zeros(0, np.uint32))
    if rank != root:
        _send_array(blk, root)
        _send_array(np.asarray(off, np.uint64), root)
        _send_array(np.asarray(info, np.uint32), root)
        _send_array(np.asarray(rec, np.uint16), root)
        return None
    out = [None] * int(n_total.item())
    infos = np.zeros(int(n_total.item()), np.uint32)
    for r in range(world):
        if r == root:
            rblk, roff, rinfo, rrec = blk, np.asarray(off, np.uint64), np.asarray(info, np.uint32), np.asarray(rec, np.uint16)
        else:
            rblk, roff, rinfo, rrec = (_recv_array(r, np.int64), _recv_array(r, np.uint64), _recv_array(r, np.uint32),
                                       _recv_array(r, np.uint16))
        for k, g in enumerate(rblk):
            out[int(g)] = rrec[int(roff[k]): int(roff[k + 1])].copy()
            infos[int(g)] = rinfo[k]
    return out, infos
