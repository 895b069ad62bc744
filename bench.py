#!/usr/bin/env python3
"""Headline benchmark: Mbins/s encode+decode of the CABAC bin codec on synthetic bin buffers
(BASELINE.json metric; workloads = SURVEY.md §8d configs, entropy_coding_amd/workload.py).

  python bench.py --gpus N --steps K --warmup W [--workload C4]
  (N > 1: either under python -m torch.distributed.run --nproc-per-node N ..., or bare — the process then starts the
  N rank processes itself before anything touches a GPU)

A *step* is one pass of the hot path over one batch: encode every substream of the batch, then decode
every substream back (two kernel launches), with all inputs already resident in HBM.  One process per
GPU; substreams are independent, so each rank codes its own shard (weak scaling: the per-GPU batch is
fixed) with no data-path collective; RCCL is used only for the barrier, the max-over-ranks reduction
and an (untimed, reported) gather of the per-substream sizes.  Rank 0 prints ONE JSON line.

The oracle / compiled reference are used here only (a) by the cpu_baseline leg and (b) to verify the
hashes after the timed region — never inside it.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
DESC_BYTES, RESULT_BYTES = 32, 8


def decode_geometry(dec_variant, n_records):
    """Substreams per wave the decode dispatch picks for this batch (mirror of launch_decode_v4 in csrc/cabac_kernels_v4.hip)."""
    forced = {4: 4, 8: 16, 1: 1}.get(dec_variant & 0xFF)
    if forced:
        return forced
    n = np.asarray(n_records, np.int64)
    n_sub, longest = len(n), int(n.max()) if len(n) else 0
    if n_sub <= 1024:
        return 1
    if n_sub <= 3072:   # decode_select_solo_kernel: the substreams longer than a sixteenth of the longest all among the first 1 024?
        where = np.nonzero(n > (longest >> 4))[0]
        return 1 if longest and len(where) and int(where[-1]) < 1024 else 4
    if n_sub >= 9216 and longest and int(n.sum()) >= 9216 * longest:
        return 16
    return 4


def kernel_names(enc_variant, dec_variant, n_sub):
    """Which kernels the library dispatches to (mirror of launch_encode/launch_decode in csrc/cabac_kernels.hip)."""
    enc = {4: "v4", 6: "v6", 7: "v7"}.get(enc_variant & 0xFF, "v7" if n_sub >= 3072 else "v6")
    return "encode_kernel_" + enc, "decode_kernel_v4"   # (both decode geometries are instances of decode_kernel_v4)


def vector_issue(workload, kernel, kernel_ms):
    """The vector instructions one launch executes (SQ counters of the committed profile run of THIS workload) against what
    the vector pipes can take.  Two rates are stated, neither of them a roof this line claims to sit on: the guide's
    (MI355X_MICROARCH.md, per-instruction cycle constants: a wave-wide VALU instruction every 4 cycles from a lone wave,
    every 2 once a SIMD holds two or more waves: 1 024 SIMDs x 2.4 GHz / 4 = 614 and / 2 = 1 229 Ginstr/s) and the highest
    rate THIS instruction mix has been seen to sustain on the chip (16 384 substreams, four waves per SIMD: 4 x 585 M
    instructions in 4.07 ms = 575 Ginstr/s, profiles/r02_batch_scaling.txt).  None without the profile."""
    if workload != "C4":
        return None
    try:
        name = next(n for n in ("r03_sq_instruction_mix.txt", "r02_sq_instruction_mix.txt") if os.path.exists(os.path.join(ROOT, "profiles", n)))
        for line in open(os.path.join(ROOT, "profiles", name)):
            if line.split("<")[0].strip() == kernel:
                valu = float(line.split("VALU")[1].split()[0])
                rate = valu / (kernel_ms * 1e-3) / 1e9
                return {"valu_instructions_per_launch": valu, "achieved_ginstr_s": round(rate, 1),
                        "peak_ginstr_s": {"lone_wave_per_simd": 614.4, "two_or_more_waves_per_simd": 1228.8},
                        "frac_of_lone_wave_rate": round(rate / 614.4, 4), "frac_of_nominal_rate": round(rate / 1228.8, 4),
                        "highest_rate_seen_with_this_mix_ginstr_s": 575.0, "frac_of_highest_seen": round(rate / 575.0, 4),
                        "source": "profiles/%s (replayed, not live)" % name}
    except (OSError, ValueError, IndexError, StopIteration):
        pass
    return None


def measured_traffic(workload, kernel):
    """HBM bytes per launch from the committed PMC runs (profiles/pmc_traffic.json), or None."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return t["workloads"][workload][kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C4", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--enc-variant", type=int, default=0)
    ap.add_argument("--dec-variant", type=int, default=0)
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: rank 0 holds ONE batch of the workload, LPT-shards it over the ranks "
                         "(sharding.scatter_substreams over RCCL), every rank codes its shard, the coded bytes are gathered "
                         "back (BASELINE config 5: --workload C5 --strong)")
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling (the default): every rank codes its own batch of the workload — with --workload C5 that is "
                         "8 192 substreams per GPU, the multi-GPU case in which every GPU has the parallelism to be busy")
    ap.add_argument("--no-residual", action="store_true", help="skip the residual-binariser leg (C4, one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-pointer (PCIe-inclusive) leg")
    ap.add_argument("--no-co-scheduled", action="store_true", help="skip the two-stream (encode beside decode) leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    return ap.parse_args()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _usable_cores():
    """Cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota where one is set
    (a GPU box gives each lease a share of the host, not all of its cores)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    if os.environ.get("CABAC_BENCH_CPU_THREADS"):
        n = int(os.environ["CABAC_BENCH_CPU_THREADS"])
    return n


def cpu_baseline(cfg, desc, records, budget_s):
    """Reference CPU path timed on this host on a bounded sample of the same workload: one substream per task on a
    native thread pool (std::thread / pthreads inside the checker library), first with 1 thread, then with every core
    this process may use.  kind 'reference' = the reference's own sources compiled by oracle/Makefile (oracle/_ref);
    kind 'port' = oracle/cabac_oracle.c when that library is not present.  `value` is the all-cores rate (`cores`
    threads); the 1-thread rate — the "10x single thread" comparator of BASELINE.json — is in `one_thread`."""
    import ctypes
    import helpers as H
    if H.ref_available():
        lib, kind, fn = H.load_ref().lib, "reference", "ref_roundtrip_mt"
    else:
        lib, kind, fn = H.load_oracle().lib, "port", "orc_roundtrip_mt"
    run = getattr(lib, fn)
    run.restype = ctypes.c_uint64
    run.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    n_sub = len(desc)
    cores = _usable_cores()
    desc = np.ascontiguousarray(desc)

    def leg(first, count, threads):
        out = np.zeros(4, np.uint64)
        wall = run(desc.ctypes.data, first, count, records.ctypes.data, threads, out.ctypes.data)
        assert int(out[1]) == 0, "CPU baseline: %d substreams did not round-trip" % int(out[1])
        return int(out[0]), wall * 1e-9, float(out[2]) * 1e-9, float(out[3]) * 1e-9

    # size the samples from a short probe so that each leg takes about budget_s / 2 of wall time
    probe = min(n_sub, 4)
    bins_p, wall_p, _, _ = leg(0, probe, 1)
    per_sub = wall_p / probe
    n1 = int(max(1, min(n_sub, (budget_s / 2) / per_sub)))
    bins1, wall1, enc1, dec1 = leg(0, n1, 1)
    # all cores: the lease's share of the host is not always visible (no cgroup quota file), so 16 threads — the share
    # of a one-GPU lease — are tried beside the full affinity mask and the faster is reported with its thread count
    best = None
    cands = sorted({min(cores, 16), cores})
    for th in cands:
        leg(0, min(n_sub, 2 * th), th)  # wake the cores up (the first threaded pass on an idle host runs far below its rate)
        nall = min(n_sub, int(max(th, (budget_s / 2 / len(cands)) / per_sub * min(th, 16))))
        bins_t, wall_t, _, _ = leg(0, nall, th)
        if best is None or bins_t / wall_t > best[0] / best[1]:
            best = (bins_t, wall_t, th, nall)
    bins_a, wall_a, cores, nall = best
    return {
        "value": round(2 * bins_a / wall_a / 1e6, 2),
        "unit": "Mbins/s",
        "cores": cores,
        "kind": kind,
        "cpu_model": _cpu_model(),
        "compiler_flags": "g++ -O2" if kind == "reference" else "gcc -O2",
        "sample": "%s substreams 0..%d (%d bins) encode+decode on %d threads, one substream per task; one_thread: substreams 0..%d (%d bins)"
                  % (cfg.name, nall - 1, bins_a, cores, n1 - 1, bins1),
        "one_thread": {"value": round(2 * bins1 / (enc1 + dec1) / 1e6, 2), "cores": 1,
                       "encode_mbins_s": round(bins1 / enc1 / 1e6, 2), "decode_mbins_s": round(bins1 / dec1 / 1e6, 2)},
    }


def whole_batch_hash(desc, records, res_e, host_bytes):
    """Every substream of the batch, not a sample: a digest of the bytes the compiled reference (oracle/_ref, or the C
    restatement where it is not built) produces for each substream against the same digest of the device's bytes.  After
    the timed region; the checker's libraries are test infrastructure."""
    import ctypes
    import helpers as H
    n = len(desc)
    desc = np.ascontiguousarray(desc)
    want = np.zeros(n, np.uint64)
    if H.ref_available():
        lib, fn, kind = H.load_ref().lib, "ref_digest_mt", "reference"
    else:
        lib, fn, kind = H.load_oracle().lib, "orc_digest_mt", "port"
    run = getattr(lib, fn)
    run.restype = ctypes.c_uint64
    run.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    t0 = time.perf_counter()
    bad = run(desc.ctypes.data, 0, n, records.ctypes.data, min(_usable_cores(), 32), want.ctypes.data)
    got = np.zeros(n, np.uint64)
    dig = H.load_oracle().lib.orc_digest_slots
    dig.restype = None
    dig.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    res = np.ascontiguousarray(res_e)
    dig(desc.ctypes.data, res.ctypes.data, n, host_bytes.ctypes.data, got.ctypes.data)
    differ = int((got != want).sum())
    return {"substreams": n, "differ": differ + int(bad), "checker": kind, "seconds": round(time.perf_counter() - t0, 2),
            "match": differ == 0 and int(bad) == 0}


def co_scheduled_leg(local_rank, variants, n_sub, n_bins, n_slots, t_desc, t_rec, bytes_total, want_bins, steps, warmup):
    """The same step — one encode and one decode of the whole resident batch — with the two kernels on two streams: the
    decode of step k (reading what the encode of step k wrote, after its event) runs beside the encode of step k + 1
    (into the other of two byte buffers, after the decode that last read it).  On the headline batch either kernel alone
    leaves issue slots of the SIMDs unused (decode: one wave per SIMD); together they fill them.  Not `value`: the
    per-kernel durations the roofline is computed from are only meaningful when a kernel has the chip to itself."""
    import torch
    from entropy_coding_amd import capi
    s_enc, s_dec = torch.cuda.Stream(), torch.cuda.Stream()
    h_enc = capi.CabacHip(local_rank, stream=s_enc.cuda_stream)
    h_dec = capi.CabacHip(local_rank, stream=s_dec.cuda_stream)
    h_enc.set_variant(*variants)
    h_dec.set_variant(*variants)
    bufs = [torch.zeros(max(bytes_total, 16), dtype=torch.uint8, device="cuda") for _ in range(2)]
    res_e = [torch.zeros(max(n_sub, 1) * 2, dtype=torch.int32, device="cuda") for _ in range(2)]
    res_d = torch.zeros(max(n_sub, 1) * 2, dtype=torch.int32, device="cuda")
    bins = torch.zeros(max(n_slots, 1), dtype=torch.uint8, device="cuda")
    coded = [torch.cuda.Event() for _ in range(2)]
    read = [torch.cuda.Event() for _ in range(2)]
    torch.cuda.synchronize()
    k = 0

    def step():
        nonlocal k
        b = k & 1
        if k >= 2:
            s_enc.wait_event(read[b])            # the decode of step k - 2 has finished with this buffer
        h_enc.encode_device(n_sub, t_desc.data_ptr(), t_rec.data_ptr(), bufs[b].data_ptr(), res_e[b].data_ptr())
        coded[b].record(s_enc)
        s_dec.wait_event(coded[b])
        h_dec.decode_device(n_sub, t_desc.data_ptr(), t_rec.data_ptr(), bufs[b].data_ptr(), bins.data_ptr(), res_d.data_ptr())
        read[b].record(s_dec)
        k += 1

    for _ in range(warmup + 2):
        step()
    torch.cuda.synchronize()
    h_enc.profile_enable(steps)
    h_dec.profile_enable(steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    enc_ms = float(np.mean([m for kk, m in h_enc.profile_read() if kk == 0]))
    dec_ms = float(np.mean([m for kk, m in h_dec.profile_read() if kk == 1]))
    ok = (not bool(res_d.view(-1, 2)[:, 1].any().item()) and not bool(res_e[0].view(-1, 2)[:, 1].any().item())
          and not bool(res_e[1].view(-1, 2)[:, 1].any().item()) and bool(torch.equal(bins[:n_slots], want_bins))
          and bool(torch.equal(bufs[0], bufs[1])))
    h_enc.close()
    h_dec.close()
    return {"mbins_s": round(2 * n_bins / (ms * 1e-3) / 1e6, 1), "ms_per_step": round(ms, 4), "steps": steps, "round_trip": ok,
            "kernel_ms_while_sharing_the_chip": {"encode": round(enc_ms, 4), "decode": round(dec_ms, 4)},
            "what": "encode and decode of the batch on two streams (decode of step k beside the encode of step k + 1, "
                    "double-buffered bytes, event-ordered); wall time per step"}


def end_to_end_leg(hip, cfg, desc, records, bytes_total, n_bins, reps=3):
    """The host-pointer path, PCIe included: cabac_hip_encode_batch / cabac_hip_decode_batch from host memory to host
    memory (chunked H2D / kernel / D2H on separate streams, compacted output, see cabac_hip.h), once with the caller's
    buffers in pinned memory (cabac_hip_host_alloc — what the shim's mirrors use under usePinnedMirrors) and once in
    pageable memory (through the library's pinned bounce ring).  Wall time of the synchronous call, best of `reps`."""
    from entropy_coding_amd import capi
    out = {}
    want_bins = (records >> 15).astype(np.uint8)
    ref_bytes = None
    for kind in ("pinned", "pageable"):
        if kind == "pinned":
            keep = [capi.PinnedArray(records.shape, np.uint16), capi.PinnedArray((bytes_total,), np.uint8),
                    capi.PinnedArray((len(records),), np.uint8)]
            h_rec, h_out, h_bins = (k.array for k in keep)
            h_rec[:] = records
        else:
            keep = []
            h_rec, h_out, h_bins = records, np.zeros(bytes_total, np.uint8), np.zeros(len(records), np.uint8)
        t_enc, t_dec = [], []
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            _, res = hip.encode_batch(desc, h_rec, bytes_total, out=h_out)
            t1 = time.perf_counter()
            t_enc.append(t1 - t0)
        # ... and with the output as a multiplexer takes it: the substreams back to back (cabac_hip_encode_batch_payload)
        nb0 = (res["n_bits"].astype(np.int64) + 7) // 8
        if kind == "pinned":
            keep.append(capi.PinnedArray((int(nb0.sum()) + 16,), np.uint8))
            h_pay = keep[-1].array
        else:
            h_pay = np.zeros(int(nb0.sum()) + 16, np.uint8)
        t_pay = []
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            offs, res_p = hip.encode_batch_payload(desc, h_rec, h_pay)
            t_pay.append(time.perf_counter() - t0)
        pay_ok = bool(np.array_equal(res_p["n_bits"], res["n_bits"])) and int(offs[-1]) == int(nb0.sum())
        for s in range(0, len(desc), max(len(desc) // 256, 1)):
            o = int(desc["byte_offset"][s])
            pay_ok = pay_ok and bool(np.array_equal(h_pay[int(offs[s]): int(offs[s + 1])], h_out[o:o + int(nb0[s])]))
        # decode input as a decoder has it: the coded substreams packed (16-byte aligned slots), not the encoder's
        # worst-case slots
        nb = (res["n_bits"].astype(np.int64) + 7) // 8
        ddesc = desc.copy()
        ddesc["byte_capacity"] = nb
        slot = (nb + 15) // 16 * 16
        ddesc["byte_offset"] = np.concatenate([[0], np.cumsum(slot)[:-1]])
        in_total = int(slot.sum())
        if kind == "pinned":
            keep.append(capi.PinnedArray((in_total,), np.uint8))
            h_in = keep[-1].array
        else:
            h_in = np.zeros(in_total, np.uint8)
        for s in range(len(desc)):
            o, q, n = int(desc["byte_offset"][s]), int(ddesc["byte_offset"][s]), int(nb[s])
            h_in[q:q + n] = h_out[o:o + n]
        for _ in range(reps + 1):
            t2 = time.perf_counter()
            _, rd = hip.decode_batch(ddesc, h_rec, h_in, bins=h_bins)
            t3 = time.perf_counter()
            t_dec.append(t3 - t2)
        ok = not res["flags"].any() and not rd["flags"].any()
        live = np.zeros(len(records), bool)
        for s in range(len(desc)):
            o, n = int(desc["rec_offset"][s]), int(desc["n_records"][s])
            live[o:o + n] = True
        ok = ok and bool(np.array_equal(h_bins[live], want_bins[live]))
        # ... and with the decoded bins packed eight to a byte (cabac_hip_decode_batch_packed): an eighth of the D2H bytes
        if kind == "pinned":
            kp = capi.PinnedArray(((len(records) + 7) // 8 + 1,), np.uint8)
            keep.append(kp)
            h_pk = kp.array
        else:
            h_pk = np.zeros((len(records) + 7) // 8 + 1, np.uint8)
        t_pk = []
        for _ in range(reps + 1):
            t2 = time.perf_counter()
            pk, rp = hip.decode_batch_packed(ddesc, h_rec, h_in, packed=h_pk)
            t_pk.append(time.perf_counter() - t2)
        ok = ok and not rp["flags"].any() and bool(np.array_equal(np.unpackbits(pk, bitorder="little")[: len(records)][live], want_bins[live]))
        coded = np.concatenate([h_out[int(desc["byte_offset"][s]): int(desc["byte_offset"][s]) + (int(res["n_bits"][s]) + 7) // 8]
                                for s in range(0, len(desc), max(len(desc) // 256, 1))])
        if ref_bytes is None:
            ref_bytes = coded.copy()
        ok = ok and bool(np.array_equal(coded, ref_bytes))
        e, d = min(t_enc[1:]), min(t_dec[1:])
        ok = ok and pay_ok
        out[kind] = {"encode_ms": round(e * 1e3, 3), "encode_payload_ms": round(min(t_pay[1:]) * 1e3, 3), "decode_ms": round(d * 1e3, 3),
                     "decode_packed_bins_ms": round(min(t_pk[1:]) * 1e3, 3),
                     "encode_mbins_s": round(n_bins / e / 1e6, 1), "decode_mbins_s": round(n_bins / d / 1e6, 1),
                     "mbins_s": round(2 * n_bins / (e + d) / 1e6, 1),
                     "mbins_s_payload_and_packed": round(2 * n_bins / (min(t_pay[1:]) + min(t_pk[1:])) / 1e6, 1), "round_trip": bool(ok)}
        for k in keep:
            k.close()
    h2d_enc = 2 * n_bins + 32 * len(desc)
    out["pcie_bytes"] = {"encode_h2d": h2d_enc, "encode_d2h": int(nb.sum()) + 16 * len(desc), "decode_h2d": h2d_enc + in_total,
                         "decode_d2h": int(len(records))}
    out["what"] = "cabac_hip_encode_batch + cabac_hip_decode_batch, host memory to host memory, wall time of the calls"
    return out


def n_coef_total(c_u, copies):
    return c_u * copies


def residual_leg(hip, n_tiles, unique=256, reps=4, host_e2e=True):
    """Coefficient blocks -> bin records (cabac_hip_residual_device), both passes, on n_tiles tiles of
    workload.RESIDUAL_TILE_MIX (`unique` generated tiles, replicated on the device).  Checked against the md5 the
    compiled reference produced for the first blocks (tests/golden/residual_bench.json) and replica against replica."""
    import torch
    from entropy_coding_amd import capi, workload as W
    unique = min(unique, n_tiles)
    copies = max(n_tiles // unique, 1)
    n_tiles = copies * unique
    tus, coeff, tile_first = W.build_residual_tiles(unique)
    n_u, c_u = len(tus), len(coeff)
    all_tus = np.tile(tus, copies)
    all_tus["coeff_offset"] += np.repeat(np.arange(copies, dtype=np.uint64) * np.uint64(c_u), n_u)
    n = len(all_tus)
    t_tu = torch.from_numpy(all_tus.view(np.uint8).reshape(-1).copy()).cuda()
    t_co = torch.from_numpy(coeff).cuda().repeat(copies)
    t_cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    t_info = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    hip.profile_enable(2 * reps + 2)
    for _ in range(reps + 1):
        hip.residual_device(n, t_tu.data_ptr(), t_co.data_ptr(), 0, t_cnt.data_ptr(), t_info.data_ptr(), 0)
    hip.synchronize()
    cnt = t_cnt.to(torch.int64)
    # one substream per tile: its blocks' records back to back, then TRM(1)
    per_tile = n_u // unique
    tile_of = torch.arange(n, device="cuda") // per_tile
    t_off = torch.cumsum(cnt, 0) - cnt + tile_of
    n_bins = int(cnt.sum().item())
    t_rec = torch.zeros(n_bins + n_tiles, dtype=torch.int16, device="cuda")
    tile_bins = cnt.view(n_tiles, per_tile).sum(1)
    sub_first = torch.cumsum(tile_bins + 1, 0) - (tile_bins + 1)
    t_rec[sub_first + tile_bins] = -32257  # 0x81FF: TRM(1)
    torch.cuda.synchronize()
    for _ in range(reps + 1):
        hip.residual_device(n, t_tu.data_ptr(), t_co.data_ptr(), t_off.data_ptr(), t_cnt.data_ptr(), t_info.data_ptr(),
                            t_rec.data_ptr())
    prof = [ms for k, ms in hip.profile_read() if k == 5]
    p1, p2 = float(np.mean(prof[1:reps + 1])), float(np.mean(prof[reps + 2:]))
    per = (n_bins + n_tiles) // copies
    ok = (n_bins + n_tiles) == per * copies and all(bool(torch.equal(t_rec[:per], t_rec[r * per:(r + 1) * per])) for r in range(1, copies))
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "residual_bench.json")))
    k = gold["n_blocks"]
    assert k <= per_tile
    first = t_rec[: int(cnt[:k].sum().item())].cpu().numpy().view(np.uint16)
    ok = ok and hashlib.md5(first.tobytes()).hexdigest() == gold["records_md5"] and not bool((t_info < 0).any().item())
    # ... and on through the bin encoder: coefficients -> bytes without leaving the device; decode gives the bins back
    desc = np.zeros(n_tiles, capi.DESC_DTYPE)
    lens = (tile_bins + 1).cpu().numpy().astype(np.uint64)
    desc["n_records"] = lens
    desc["rec_offset"] = sub_first.cpu().numpy().astype(np.uint64)
    cap = ((lens * 3) // 4 + 64 + 15) // 16 * 16
    desc["byte_capacity"] = cap
    desc["byte_offset"] = np.concatenate([[0], np.cumsum(cap)[:-1]])
    desc["qp"] = 32
    desc["init_id"] = 2 | capi.SUB_FINISH | capi.SUB_ALIGN_RBSP
    t_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).cuda()
    t_bytes = torch.zeros(int(cap.sum()), dtype=torch.uint8, device="cuda")
    t_res = torch.zeros(2 * n_tiles, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    hip.profile_enable(reps + 1)
    for _ in range(reps + 1):
        hip.encode_device(n_tiles, t_desc.data_ptr(), t_rec.data_ptr(), t_bytes.data_ptr(), t_res.data_ptr())
    enc = float(np.mean([ms for kk, ms in hip.profile_read() if kk == 0][1:]))
    res = t_res.cpu().numpy().view(capi.RESULT_DTYPE)
    ddesc = desc.copy()
    ddesc["byte_capacity"] = (res["n_bits"] + 7) // 8
    t_ddesc = torch.from_numpy(ddesc.view(np.uint8).reshape(-1).copy()).cuda()
    t_bins = torch.zeros(n_bins + n_tiles, dtype=torch.uint8, device="cuda")
    t_res_d = torch.zeros(2 * n_tiles, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    hip.profile_enable(reps + 1)
    for _ in range(reps + 1):
        hip.decode_device(n_tiles, t_ddesc.data_ptr(), t_rec.data_ptr(), t_bytes.data_ptr(), t_bins.data_ptr(), t_res_d.data_ptr())
    hip.synchronize()
    dec = float(np.mean([ms for kk, ms in hip.profile_read() if kk == 1][1:]))
    round_trip = (not res["flags"].any() and not t_res_d.cpu().numpy().view(capi.RESULT_DTYPE)["flags"].any()
                  and bool(torch.equal(t_bins, (t_rec < 0).to(torch.uint8))))
    ok = ok and round_trip
    out_bytes = int(((res["n_bits"].astype(np.int64) + 7) // 8).sum())
    # ... the same in ONE call with the block records spliced into the substreams on the device (cabac_hip_encode_residual_device:
    # sizes pass -> prefix sums -> records pass straight into the expanded substreams -> encode -> compaction): the host side
    # of a tile is its terminate bin and 400 splices in front of it.  Wall time of the call (it waits once in the middle).
    sp_desc = np.zeros(n_tiles, capi.DESC_DTYPE)
    sp_desc["n_records"], sp_desc["rec_offset"], sp_desc["qp"] = 1, np.arange(n_tiles, dtype=np.uint64), 32
    sp_desc["init_id"] = 2 | capi.SUB_FINISH | capi.SUB_ALIGN_RBSP
    sp_first = (np.arange(n_tiles + 1, dtype=np.uint32) * per_tile).astype(np.uint32)
    sp_list = np.zeros(n, capi.SPLICE_DTYPE)
    sp_list["tu"] = np.arange(n, dtype=np.uint32)
    sp_rec = np.full(n_tiles, 0x81FF, np.uint16)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
    t_sp_desc, t_sp_first, t_sp_list, t_sp_rec = dev(sp_desc), dev(sp_first), dev(sp_list), dev(sp_rec)
    t_pay = torch.zeros(out_bytes + 64, dtype=torch.uint8, device="cuda")
    t_poff = torch.zeros(n_tiles + 1, dtype=torch.int64, device="cuda")
    t_pres = torch.zeros(2 * n_tiles, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    t_sp = []
    hip.profile_enable(0)
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        hip.encode_residual_device(n_tiles, t_sp_desc.data_ptr(), t_sp_rec.data_ptr(), t_sp_first.data_ptr(), t_sp_list.data_ptr(), n, n,
                                   t_tu.data_ptr(), t_co.data_ptr(), t_pay.data_ptr(), out_bytes + 64, t_poff.data_ptr(), t_pres.data_ptr())
        hip.synchronize()
        t_sp.append(time.perf_counter() - t0)
    hip.profile_enable(16)
    hip.encode_residual_device(n_tiles, t_sp_desc.data_ptr(), t_sp_rec.data_ptr(), t_sp_first.data_ptr(), t_sp_list.data_ptr(), n, n,
                               t_tu.data_ptr(), t_co.data_ptr(), t_pay.data_ptr(), out_bytes + 64, t_poff.data_ptr(), t_pres.data_ptr())
    sp_prof = {}
    for kk, ms in hip.profile_read():
        sp_prof[kk] = sp_prof.get(kk, 0.0) + ms
    nb_all = (res["n_bits"].astype(np.int64) + 7) // 8
    pres = t_pres.cpu().numpy().view(capi.RESULT_DTYPE)
    poff = t_poff.cpu().numpy()
    spliced_ok = bool(np.array_equal(pres["n_bits"], res["n_bits"])) and not pres["flags"].any() and int(poff[-1]) == out_bytes
    host_pay, host_slots = t_pay.cpu().numpy(), t_bytes.cpu().numpy()
    for s in range(0, n_tiles, max(n_tiles // 512, 1)):
        o = int(desc["byte_offset"][s])
        spliced_ok = spliced_ok and bool(np.array_equal(host_pay[int(poff[s]): int(poff[s + 1])], host_slots[o:o + int(nb_all[s])]))
    spliced = {"call_ms": round(min(t_sp[1:]) * 1e3, 4), "mcoeff_s": round(c_u * copies / min(t_sp[1:]) / 1e6, 1),
               "mbins_s": round(n_bins / min(t_sp[1:]) / 1e6, 1),
               "kernel_ms": {"residual_passes": round(sp_prof.get(5, 0.0), 4), "splice_plan_scan_expand": round(sp_prof.get(10, 0.0), 4),
                             "encode": round(sp_prof.get(0, 0.0), 4), "assemble": round(sp_prof.get(6, 0.0), 4)},
               "bytes_match_record_path": bool(spliced_ok),
               "what": "cabac_hip_encode_residual_device: coefficients + 1 host record per tile -> compacted coded substreams, wall time"}
    if host_e2e:
        # ... and from pinned host memory (cabac_hip_encode_batch_residual): 4 bytes per coefficient go up, only bytes come back
        keep = [capi.PinnedArray((c_u * copies,), np.int32), capi.PinnedArray((out_bytes + 64,), np.uint8)]
        for r in range(copies):
            keep[0].array[r * c_u:(r + 1) * c_u] = coeff
        t_h = []
        for _ in range(3):
            t0 = time.perf_counter()
            h_off, h_res = hip.encode_batch_residual(sp_desc, sp_rec, sp_first, sp_list, all_tus, keep[0].array, keep[1].array)
            t_h.append(time.perf_counter() - t0)
        e2e_ok = bool(np.array_equal(h_res["n_bits"], res["n_bits"])) and bool(np.array_equal(keep[1].array[:out_bytes], host_pay[:out_bytes]))
        spliced["from_pinned_host"] = {"call_ms": round(min(t_h[1:]) * 1e3, 3), "mcoeff_s": round(c_u * copies / min(t_h[1:]) / 1e6, 1),
                                       "mbins_s": round(n_bins / min(t_h[1:]) / 1e6, 1), "h2d_bytes": int(4 * c_u * copies + 24 * n + 38 * n_tiles),
                                       "d2h_bytes": int(out_bytes + 16 * n_tiles), "bytes_match": e2e_ok}
        spliced_ok = spliced_ok and e2e_ok
        # ... and with the coefficients as int16 (cabac_hip_encode_batch_residual16; what the C++ shim's staging area hands over when
        # the blocks' dynamic range is 15 bits): 2 bytes per coefficient go up
        if int(coeff.min()) >= -32768 and int(coeff.max()) <= 32767:
            keep.append(capi.PinnedArray((c_u * copies,), np.int16))
            for r in range(copies):
                keep[2].array[r * c_u:(r + 1) * c_u] = coeff
            t_h = []
            for _ in range(3):
                keep[1].array[:] = 0
                t0 = time.perf_counter()
                h_off, h_res = hip.encode_batch_residual(sp_desc, sp_rec, sp_first, sp_list, all_tus, keep[2].array, keep[1].array)
                t_h.append(time.perf_counter() - t0)
            ok16 = bool(np.array_equal(h_res["n_bits"], res["n_bits"])) and bool(np.array_equal(keep[1].array[:out_bytes], host_pay[:out_bytes]))
            spliced["from_pinned_host_int16"] = {"call_ms": round(min(t_h[1:]) * 1e3, 3), "mcoeff_s": round(c_u * copies / min(t_h[1:]) / 1e6, 1),
                                                 "mbins_s": round(n_bins / min(t_h[1:]) / 1e6, 1),
                                                 "h2d_bytes": int(2 * c_u * copies + 24 * n + 38 * n_tiles), "bytes_match": ok16}
            spliced_ok = spliced_ok and ok16
        for kp in keep:
            kp.close()
    ok = ok and spliced_ok
    # ... and back: the residual parser turns the bytes into coefficient blocks again, deriving every context itself.
    # The workload codes with sign-data hiding and its hidden signs are arranged as an encoder's quantiser leaves them
    # (workload._arrange_hidden_signs), so every coefficient must come back exactly.
    t_first = (torch.arange(n_tiles + 1, device="cuda", dtype=torch.int32) * per_tile).contiguous()
    t_dec = torch.zeros_like(t_co)
    t_res_p = torch.zeros(2 * n_tiles, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    hip.profile_enable(reps + 1)
    for _ in range(reps + 1):
        hip.residual_parse_device(n_tiles, t_ddesc.data_ptr(), t_bytes.data_ptr(), t_first.data_ptr(), t_tu.data_ptr(),
                                  t_dec.data_ptr(), t_res_p.data_ptr())
    parse_ms = float(np.mean([ms for kk, ms in hip.profile_read() if kk == 9][1:]))
    res_p = t_res_p.cpu().numpy().view(capi.RESULT_DTYPE)
    parsed_back = (not res_p["flags"].any() and bool(np.array_equal(res_p["n_bits"], t_res_d.cpu().numpy().view(capi.RESULT_DTYPE)["n_bits"]))
                   and bool(torch.equal(t_dec, t_co)))
    bytes_p = out_bytes + 4 * n_coef_total(c_u, copies) + (16 + 8) * n + 40 * n_tiles
    ok = ok and parsed_back
    parse_host = None
    if host_e2e:
        # ... and host memory to host memory (cabac_hip_residual_parse_batch / _batch16): the bytes go up, the blocks come back —
        # as int32 (with the caller's buffer uploaded first: what the parser does not write keeps its values) and as int16
        h_in = t_bytes.cpu().numpy()
        h_first = (np.arange(n_tiles + 1, dtype=np.int64) * per_tile).astype(np.uint32)
        want_co = t_co.cpu().numpy()
        parse_host = {}
        for name, dt in (("int32", np.int32), ("int16", np.int16)):
            if dt == np.int16 and not (int(want_co.min()) >= -32768 and int(want_co.max()) <= 32767):
                continue
            kp = capi.PinnedArray((c_u * copies,), dt)
            t_h = []
            for _ in range(3):
                kp.array[:] = 0
                t0 = time.perf_counter()
                co_h, res_h = hip.residual_parse_batch(ddesc, h_in, h_first, all_tus, c_u * copies, int16=(dt == np.int16), coeff=kp.array)
                t_h.append(time.perf_counter() - t0)
            okh = not res_h["flags"].any() and bool(np.array_equal(co_h.astype(np.int32), want_co))
            parse_host[name] = {"call_ms": round(min(t_h[1:]) * 1e3, 3), "mcoeff_s": round(c_u * copies / min(t_h[1:]) / 1e6, 1),
                                "d2h_bytes": int(dt().itemsize * c_u * copies), "coefficients_match": bool(okh)}
            ok = ok and okh
            kp.close()
    n_coef = c_u * copies
    bytes1 = 4 * n_coef + (16 + 4 + 4) * n
    bytes2 = 4 * n_coef + 2 * n_bins + (16 + 8 + 4 + 4) * n
    return {"kernel": "residual_kernel", "workload": "%d tiles x %d transform blocks (1080p/64 tile mix), %d unique tiles" % (n_tiles, n_u // unique, unique),
            "blocks": n, "coefficients": n_coef, "bins": n_bins,
            "kernel_ms": {"pass1_count": round(p1, 4), "pass2_write": round(p2, 4)},
            "mcoeff_s": round(n_coef / ((p1 + p2) * 1e-3) / 1e6, 1), "mbins_s": round(n_bins / ((p1 + p2) * 1e-3) / 1e6, 1),
            "algorithmic_bytes_per_launch": {"pass1_count": bytes1, "pass2_write": bytes2},
            "achieved_gbps": {"pass1_count": round(bytes1 / (p1 * 1e-3) / 1e9, 2), "pass2_write": round(bytes2 / (p2 * 1e-3) / 1e9, 2)},
            "frac_of_hbm_peak": round(bytes2 / (p2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
            # PMC bytes per launch of residual_kernel<false> / <true> (profiles/pmc_traffic.json), the ordering pre-pass apart
            "traffic": {"pass1_count": measured_traffic("C4", "residual_kernel_count"),
                        "pass2_write": measured_traffic("C4", "residual_kernel_write")},
            # the same records through the bin encoder (one substream per tile, TRM-terminated), decoded back
            # (real residual records: 90 % context coded, 84 % of the bins followed within their 16 by a bin of the same
            # context in one of the wave's four substreams — the headline's synthetic mix has 5 % — and ragged substreams)
            "to_bytes": {"encode_kernel_ms": round(enc, 4), "decode_kernel_ms": round(dec, 4),
                         "mbins_s": round(2 * n_bins / ((enc + dec) * 1e-3) / 1e6, 1),
                         "coefficients_to_bytes_ms": round(p1 + p2 + enc, 4),
                         "bitstream_bytes": out_bytes, "round_trip": bool(round_trip)},
            "coefficients_to_bytes": spliced,
            # bytes -> coefficients by the residual parser (contexts derived on the device, nothing supplied but block sizes)
            "parse": {"kernel": "residual_parse_kernel", "kernel_ms": round(parse_ms, 4),
                      "mbins_s": round(n_bins / (parse_ms * 1e-3) / 1e6, 1), "mcoeff_s": round(n_coef / (parse_ms * 1e-3) / 1e6, 1),
                      "algorithmic_bytes_per_launch": bytes_p, "achieved_gbps": round(bytes_p / (parse_ms * 1e-3) / 1e9, 2),
                      "frac_of_hbm_peak": round(bytes_p / (parse_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                      "coefficients_match": bool(parsed_back), "from_host": parse_host},
            "records_match_reference": bool(ok)}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: this process — which has not imported torch and never touches a
    GPU — starts the N rank processes itself (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, as torch.distributed.run would set them), passes rank 0's JSON line through and exits non-zero if any
    rank fails."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this host driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, pending = 0, set(range(n))
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in pending:        # a failed rank leaves the others waiting in a collective: stop exactly those
                    procs[q].terminate()
        time.sleep(0.05)
    sys.exit(rc)


def main():
    args = parse_args()
    if args.weak and args.strong:
        raise SystemExit("--weak and --strong exclude each other")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))

    import torch
    import torch.distributed as dist
    from entropy_coding_amd import capi
    from entropy_coding_amd.workload import CONFIGS, build_batch

    # Rehearsal knobs for a 1-GPU box (never set by the driver): CABAC_BENCH_BACKEND=gloo runs the
    # collectives on CPU tensors, CABAC_BENCH_SAME_DEVICE=1 puts every rank on cuda:0.
    backend = os.environ.get("CABAC_BENCH_BACKEND", "nccl")
    if os.environ.get("CABAC_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    coll_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    # CABAC_BENCH_DIST_ONE=1 (rehearsal, 1-GPU box): a process group of ONE rank, so that every collective and the scatter / gather of
    # the multi-GPU path run through RCCL on device tensors (tests/test_gpu_rccl_one_rank.py)
    multi = world > 1 or os.environ.get("CABAC_BENCH_DIST_ONE") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    cfg = CONFIGS[args.workload]
    stream = torch.cuda.current_stream()
    hip = capi.CabacHip(local_rank, stream=stream.cuda_stream)
    hip.set_variant(args.enc_variant, args.dec_variant)
    scatter_ms = gather_ms_strong = None
    if args.strong:
        # ---- strong scaling: ONE batch on rank 0, sharded over the ranks ------------------------------------------
        from entropy_coding_amd import sharding
        whole = None
        if rank == 0:
            wdesc, wrec, _ = build_batch(cfg, stagger=False)
            worder = build_batch.last_order
            whole = (wdesc, torch.from_numpy(wrec.view(np.int16)).cuda())       # the batch resident on the ingest GPU
            del wrec
        if multi:
            barrier_sync = lambda: (torch.cuda.synchronize(), dist.barrier())
            barrier_sync()
            t0 = time.perf_counter()
            desc, rec_u8, bytes_total, my_idx = sharding.scatter_substreams(whole[0] if rank == 0 else None,
                                                                            whole[1] if rank == 0 else None, root=0)
            barrier_sync()
            scatter_ms = (time.perf_counter() - t0) * 1e3
            t_rec = rec_u8.view(torch.int16) if rec_u8.device.type == "cuda" else rec_u8.view(torch.int16).cuda()
        else:
            desc, t_rec, bytes_total = sharding.pack_shard(whole[0], whole[1], np.arange(len(whole[0])))
            my_idx = np.arange(len(desc), dtype=np.int64)
        n_sub = len(desc)
        order = None
        n_bins = int(desc["n_records"].astype(np.int64).sum())
        n_slots = int(t_rec.numel())
        records = None
    else:
        n_sub = cfg.n_substreams
        first = rank * n_sub  # weak scaling: rank r codes substreams [r*n_sub, (r+1)*n_sub) of the same generator
        desc, records, bytes_total = build_batch(cfg, first=first, count=n_sub)
        order = build_batch.last_order  # generator index of each descriptor row (LPT order for mixed lengths)
        n_bins = int(desc["n_records"].astype(np.int64).sum())
        n_slots = int(len(records))  # records/bins buffers include the stagger gaps (never touched)
        t_rec = torch.from_numpy(records.view(np.int16)).cuda()

    t_desc = torch.from_numpy(desc.view(np.uint8)).cuda()
    t_bytes = torch.zeros(max(bytes_total, 16), dtype=torch.uint8, device="cuda")
    t_res_e = torch.zeros(max(n_sub, 1) * 2, dtype=torch.int32, device="cuda")
    t_res_d = torch.zeros(max(n_sub, 1) * 2, dtype=torch.int32, device="cuda")
    t_bins = torch.zeros(max(n_slots, 1), dtype=torch.uint8, device="cuda")

    def step():
        hip.encode_device(n_sub, t_desc.data_ptr(), t_rec.data_ptr(), t_bytes.data_ptr(), t_res_e.data_ptr())
        hip.decode_device(n_sub, t_desc.data_ptr(), t_rec.data_ptr(), t_bytes.data_ptr(), t_bins.data_ptr(),
                          t_res_d.data_ptr())

    def barrier():
        if multi:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    hip.profile_enable(2 * args.steps)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-launch kernel durations from HIP events recorded on the launch stream during the timed region
    prof = hip.profile_read()
    enc_ms = [ms for k, ms in prof if k == 0]
    dec_ms = [ms for k, ms in prof if k == 1]
    enc_avg, dec_avg = float(np.mean(enc_ms)), float(np.mean(dec_ms))

    # ---- verification after the timed region: hashes + round trip ------------------------------
    res_e = t_res_e.cpu().numpy().view(capi.RESULT_DTYPE)[:n_sub]
    res_d = t_res_d.cpu().numpy().view(capi.RESULT_DTYPE)[:n_sub]
    ok = not res_e["flags"].any() and not res_d["flags"].any()
    want_bins = (t_rec < 0).to(torch.uint8)
    ok = ok and bool(torch.equal(t_bins[:n_slots], want_bins))
    out_bytes = int(((res_e["n_bits"].astype(np.int64) + 7) // 8).sum())
    hash_match = None
    whole_hash = None
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "synth_md5.json")))[cfg.name]
    if args.strong:
        # every rank compacts its coded substreams on the device; they travel back to rank 0 device to device and are put
        # in global substream order there: md5s of the reference's bytes (tests/golden/synth_md5.json)
        from entropy_coding_amd import sharding
        t_pay = torch.zeros(max(out_bytes, 1), dtype=torch.uint8, device="cuda")
        t_offs = torch.zeros(n_sub + 1, dtype=torch.int64, device="cuda")
        hip.assemble_device(n_sub, t_desc.data_ptr(), t_res_e.data_ptr(), t_bytes.data_ptr(), t_pay.data_ptr(), out_bytes, t_offs.data_ptr())
        hip.synchronize()
        if multi:
            torch.cuda.synchronize()
            dist.barrier()
            g0 = time.perf_counter()
            got = sharding.gather_payloads(my_idx, res_e, t_pay[:out_bytes] if backend == "nccl" else t_pay[:out_bytes].cpu(), root=0)
            torch.cuda.synchronize()
            dist.barrier()
            gather_ms_strong = (time.perf_counter() - g0) * 1e3
        else:
            got = [(my_idx, res_e, t_pay[:out_bytes])]
        if rank == 0:
            streams, _ = sharding.ordered_streams(len(whole[0]), got)
            row_of = {int(idx): k for k, idx in enumerate(worder)}
            hash_match = all(s is not None for s in streams)
            cat = hashlib.md5()
            for g in gold["substreams"]:
                b = streams[row_of[g["index"]]]
                hash_match = hash_match and hashlib.md5(b.tobytes()).hexdigest() == g["md5"]
                cat.update(b.tobytes())
            hash_match = hash_match and cat.hexdigest() == gold["concat_md5"]
            total_payload = int(sum(len(s) for s in streams))
            del streams
        # ... and every substream of every rank's shard against the checker's bytes (the golden md5s cover a sample)
        whole_hash = whole_batch_hash(desc, t_rec.cpu().numpy().view(np.uint16), res_e, t_bytes.cpu().numpy())
        if multi:
            t = torch.tensor([whole_hash["substreams"], whole_hash["differ"]], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(t)
            whole_hash.update(substreams=int(t[0].item()), differ=int(t[1].item()), match=int(t[1].item()) == 0)
        if rank == 0:
            hash_match = hash_match and whole_hash["match"]
    else:
        host_bytes = t_bytes.cpu().numpy()
        if rank == 0:
            hash_match = True
            row_of = {int(idx): k for k, idx in enumerate(order)}
            for g in gold["substreams"]:
                s = row_of.get(g["index"])
                if s is not None:
                    o, nb = int(desc["byte_offset"][s]), (int(res_e["n_bits"][s]) + 7) // 8
                    hash_match = hash_match and hashlib.md5(host_bytes[o:o + nb].tobytes()).hexdigest() == g["md5"]
        # ... and every substream of every rank's batch against the checker's bytes (the golden md5s above cover a sample)
        whole_hash = whole_batch_hash(desc, records, res_e, host_bytes)
        if multi:
            t = torch.tensor([whole_hash["substreams"], whole_hash["differ"]], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(t)
            whole_hash.update(substreams=int(t[0].item()), differ=int(t[1].item()), match=int(t[1].item()) == 0)
        if rank == 0:
            hash_match = hash_match and whole_hash["match"]
        del host_bytes

    # ---- bit estimator on the same resident records, outside the timed region ----
    t_est = torch.zeros(max(n_sub, 1), dtype=torch.int64, device="cuda")
    t_est_flags = torch.zeros(max(n_sub, 1), dtype=torch.int32, device="cuda")
    hip.estimate_device(n_sub, t_desc.data_ptr(), t_rec.data_ptr(), t_est.data_ptr(), t_est_flags.data_ptr())
    torch.cuda.synchronize()
    hip.profile_enable(4)
    for _ in range(4):
        hip.estimate_device(n_sub, t_desc.data_ptr(), t_rec.data_ptr(), t_est.data_ptr(), t_est_flags.data_ptr())
    est_ms = float(np.mean([ms for k, ms in hip.profile_read() if k == 4]))
    est_ok = not bool(t_est_flags.any().item())
    est_bits_per_bin = float(t_est.sum().item()) / 32768.0 / max(n_bins, 1)

    # ---- substream assembly / extraction / emulation count on the coded batch, untimed ----
    try:
        t_pay = torch.zeros(max(out_bytes, 1), dtype=torch.uint8, device="cuda")
        t_offs = torch.zeros(n_sub + 1, dtype=torch.int64, device="cuda")
        t_emu = torch.zeros(max(n_sub, 1), dtype=torch.int32, device="cuda")
        t_back = torch.zeros_like(t_bytes)
        torch.cuda.synchronize()
        hip.profile_enable(12)
        for _ in range(4):
            hip.assemble_device(n_sub, t_desc.data_ptr(), t_res_e.data_ptr(), t_bytes.data_ptr(), t_pay.data_ptr(), out_bytes, t_offs.data_ptr())
            hip.split_device(n_sub, t_desc.data_ptr(), t_offs.data_ptr(), t_pay.data_ptr(), t_back.data_ptr())
            hip.count_emulations_device(n_sub, t_desc.data_ptr(), t_res_e.data_ptr(), t_bytes.data_ptr(), t_emu.data_ptr())
        asm_prof = hip.profile_read()
        asm_ms = {name: float(np.mean([ms for k, ms in asm_prof if k == kind][1:])) for name, kind in (("assemble", 6), ("split", 7), ("count_emulations", 8))}
        sizes_np = (res_e["n_bits"].astype(np.int64) + 7) // 8
        asm_ok = bool(np.array_equal(t_offs.cpu().numpy(), np.concatenate([[0], np.cumsum(sizes_np)])))
        for s_ in range(0, n_sub, max(n_sub // 64, 1)):   # extraction gives every sampled substream back
            o_, nb_ = int(desc["byte_offset"][s_]), int(sizes_np[s_])
            asm_ok = asm_ok and bool(torch.equal(t_bytes[o_:o_ + nb_], t_back[o_:o_ + nb_]))
        assemble = {"kernel_ms": {k_: round(v_, 4) for k_, v_ in asm_ms.items()}, "payload_bytes": out_bytes,
                    "assemble_gbps": round(2 * out_bytes / (asm_ms["assemble"] * 1e-3) / 1e9, 2),
                    "emulations": int(t_emu.sum().item()), "round_trip": asm_ok}
        del t_back, t_pay
    except Exception as e:  # the headline line must not depend on this leg
        assemble = {"error": "%s: %s" % (type(e).__name__, e), "round_trip": False}

    # ---- residual binariser on coefficient blocks, outside the timed region ----------
    residual = None
    if world == 1 and cfg.name == "C4" and not args.no_residual and not args.strong:
        try:
            residual = residual_leg(hip, n_sub, host_e2e=not args.no_end_to_end)
        except Exception as e:  # the headline line must not depend on this leg
            residual = {"error": "%s: %s" % (type(e).__name__, e), "records_match_reference": False}

    # ---- untimed gather of the per-substream sizes over RCCL (the only exchange the path has) ---
    gather_ms = None
    if multi:
        if not args.strong:
            sizes = t_res_e.view(-1, 2)[:, 0].contiguous().to(coll_dev)
            allsz = [torch.empty_like(sizes) for _ in range(world)]
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            dist.all_gather(allsz, sizes)
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - g0) * 1e3
        flag = torch.tensor([1 if ok else 0], device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())

    bins_all = n_bins * world
    if args.strong and multi:
        t = torch.tensor([n_bins], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(t)
        bins_all = int(t.item())
    if rank == 0:
        total_bins_per_step = 2 * bins_all  # N encoded + N decoded over all ranks
        ms_per_step = elapsed / args.steps * 1e3
        value = total_bins_per_step / (elapsed / args.steps) / 1e6
        bytes_enc = 2 * n_bins + out_bytes + (DESC_BYTES + RESULT_BYTES) * n_sub
        bytes_dec = 3 * n_bins + out_bytes + (DESC_BYTES + RESULT_BYTES) * n_sub
        enc_gbps = bytes_enc / (enc_avg * 1e-3) / 1e9
        dec_gbps = bytes_dec / (dec_avg * 1e-3) / 1e9
        dominant = "decode" if dec_avg >= enc_avg else "encode"
        ach = dec_gbps if dominant == "decode" else enc_gbps
        k_enc, k_dec = kernel_names(args.enc_variant, args.dec_variant, n_sub)
        k_dom = k_dec if dominant == "decode" else k_enc
        line = {
            "metric": "Mbins/s encode+decode, intra QP%d synthetic bin buffers" % cfg.substream(0)[2],
            "value": round(value, 2),
            "unit": "Mbins/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %s" % (cfg.name, cfg.baseline_config),
                "substreams_per_gpu": n_sub,
                "bins_per_substream": cfg.substream(0)[0] if cfg.b is None else [cfg.a[0], cfg.b[0]],
                "bins_per_gpu": n_bins,
                "ctx_permille": cfg.substream(0)[1],
                "bins_per_step": total_bins_per_step,
                "sharding": ("one batch on rank 0, LPT-sharded over the ranks (send/recv over RCCL), coded bytes gathered back; no data-path collective"
                             if args.strong else "substreams split across ranks, no data-path collective"),
                "kernel_variants": {"encode": args.enc_variant, "decode": args.dec_variant},
            },
            # what this rank's batch offers the chip: the quad decoder puts four substreams in a wave (1 024 SIMDs take one wave
            # each), the sixteen-per-wave decoder is dispatched from 9 216 about equally long substreams; `equal_substreams` is
            # the batch's bins over its longest substream's — the number that decides how busy the chip can be kept
            "occupancy": {"substreams": n_sub, "equal_substreams": round(n_bins / max(int(desc["n_records"].max()), 1), 1),
                          "decode_substreams_per_wave": decode_geometry(args.dec_variant, desc["n_records"]),
                          "decode_waves_per_simd": round(-(-n_sub // decode_geometry(args.dec_variant, desc["n_records"])) / 1024.0, 3),
                          "quad_decode_waves_per_simd": round(((n_sub + 3) // 4) / 1024.0, 3),
                          "long_chain_waves_per_simd": round(((n_bins // max(int(desc["n_records"].max()), 1) + 3) // 4) / 1024.0, 3)},
            "encode_mbins_s": round(n_bins / (enc_avg * 1e-3) / 1e6, 2),
            "decode_mbins_s": round(n_bins / (dec_avg * 1e-3) / 1e6, 2),
            "kernel_ms": {"encode": round(enc_avg, 4), "decode": round(dec_avg, 4)},
            "bitstream_bytes_per_gpu": out_bytes,
            "hash_match": bool(hash_match and ok),
            "hash_whole_batch": whole_hash,
            "roofline": {
                "bound": "hbm",        # the roof the tier prices against; what limits these kernels is in `limiter`
                "limiter": "instruction issue on the serial per-substream chain (one wave per SIMD, one vector instruction per ~2 ns: ~640 per 16-bin decode step for four substreams); HBM traffic equals the algorithmic bytes",
                "kernel": k_dom,
                "vector_issue": vector_issue(cfg.name, k_dom, dec_avg if dominant == "decode" else enc_avg),
                "achieved": round(ach, 3),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBPS, 6),
                "traffic": measured_traffic(cfg.name, k_dom),  # PMC bytes per launch (profiles/pmc_traffic.json)
                "algorithmic_bytes_per_launch": bytes_dec if dominant == "decode" else bytes_enc,
                "other_kernel": {"kernel": k_enc if dominant == "decode" else k_dec,
                                 "achieved": round(enc_gbps if dominant == "decode" else dec_gbps, 3),
                                 "algorithmic_bytes_per_launch": bytes_enc if dominant == "decode" else bytes_dec,
                                 "traffic": measured_traffic(cfg.name, k_enc if dominant == "decode" else k_dec)},
            },
        }
        line["estimate"] = {  # not part of `value`: BitEstimator_Std over the same bin strings
            "kernel": "estimate_kernel", "kernel_ms": round(est_ms, 4),
            "mbins_s": round(n_bins / (est_ms * 1e-3) / 1e6, 2),
            "algorithmic_bytes_per_launch": 2 * n_bins + 48 * n_sub,
            "achieved_gbps": round((2 * n_bins + 48 * n_sub) / (est_ms * 1e-3) / 1e9, 3),
            "estimated_bits_per_bin": round(est_bits_per_bin, 5),
            "coded_bits_per_bin": round(8.0 * out_bytes / max(n_bins, 1), 5), "flags_clear": est_ok}
        line["assemble"] = assemble  # not part of `value`: addSubstream / extractSubstream / countStartCodeEmulations on the device
        if residual is not None:
            line["residual"] = residual
        if gather_ms is not None:
            line["sizes_allgather_ms"] = round(gather_ms, 3)
        if args.strong:
            line["strong"] = {"substreams_total": int(len(whole[0])), "bins_total": bins_all,
                              "scatter_ms": None if scatter_ms is None else round(scatter_ms, 3),
                              "gather_ms": None if gather_ms_strong is None else round(gather_ms_strong, 3),
                              "scatter_bytes": 2 * bins_all, "gathered_payload_bytes": total_payload,
                              "waves_per_simd": {"this_rank": round(-(-n_sub // decode_geometry(args.dec_variant, desc["n_records"])) / 1024.0, 3),
                                                 "long_chains": round(n_bins / max(int(desc["n_records"].max()), 1) / decode_geometry(args.dec_variant, desc["n_records"]) / 1024.0, 3),
                                                 "decode_substreams_per_wave": decode_geometry(args.dec_variant, desc["n_records"])},
                              "what": "scatter / gather are outside the timed region: the step is encode + decode of the resident shard"}
        if world == 1 and not args.no_co_scheduled and not args.strong:
            try:
                line["co_scheduled"] = co_scheduled_leg(local_rank, (args.enc_variant, args.dec_variant), n_sub, n_bins, n_slots,
                                                        t_desc, t_rec, bytes_total, want_bins, args.steps, args.warmup)
            except Exception as e:
                line["co_scheduled"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_end_to_end and not args.strong:
            try:
                line["end_to_end"] = end_to_end_leg(hip, cfg, desc, records, bytes_total, n_bins)
            except Exception as e:  # the headline line must not depend on this leg
                line["end_to_end"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_cpu_baseline and not args.strong:
            line["cpu_baseline"] = cpu_baseline(cfg, desc, records, args.cpu_seconds)
        if not line["hash_match"]:
            line["error"] = "bitstream hash / round-trip mismatch"
        # a failure of the untimed legs is reported inside their own objects (`residual`, `assemble`), not here
        print(json.dumps(line))
    hip.close()
    if multi:
        dist.destroy_process_group()
    if rank == 0 and not (hash_match and ok):
        sys.exit(1)


if __name__ == "__main__":
    main()
