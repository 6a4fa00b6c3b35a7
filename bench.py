#!/usr/bin/env python3
"""bench.py -- chained anchors/s of the device half of minimap2's chaining (reference chain.c:218-327)
on N MI355X, one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[3], "ava-ont self-overlap, 100k synthetic 10 kb reads, read-sharded
across 8 GPUs"): every rank holds the same-size shard of that job, 12,500 reads (~6.1k anchors per
read), generated on the host with the rank's read indices and the job's seed, uploaded to HBM
BEFORE the clock starts.  A step = one pass of the whole device half over the resident shard:
prepass (unit split, avg_qspan sums) + chain DP (f/p/v) + compaction into new_seed[], all HIP
kernels, results left in HBM.  Reads are independent, so there is no collective on the data path
(weak scaling); torch.distributed (RCCL) only provides the barrier and the max-over-ranks time.

One JSON line on rank 0: metric/value/unit as BASELINE.json names them, plus
  roofline     -- the chain-DP kernel against the HBM roofline: 24 algorithmic bytes per anchor
                  (16 B mm128_t read + 4 B f + 4 B p written) over its average launch
                  duration, measured with HIP events on the kernel's own stream.
  cpu_baseline -- the reference's mm_chain_dp_fpga (oracle/_ref, compiled from /root/reference in the
                  build container) or, if that file is absent, the oracle's port of it, timed on all
                  host cores over a bounded sample of the same batch (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READS_PER_GPU = 12_500          # 100k reads / 8 GPUs (configs[3])
SEED = 20261004
ALGO_BYTES_PER_ANCHOR = 24
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads-per-gpu", type=int, default=READS_PER_GPU)
    ap.add_argument("--preset", default="ava-ont", help="generator + DP preset (ava-ont, map-ont, skew, dense)")
    ap.add_argument("--ring", type=int, default=0, help="LDS ring capacity override (128/256/512)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse several ranks on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed side measurements (end-to-end rate, copy bandwidth, pair evaluations)")
    ap.add_argument("--cpu-sample-anchors", type=int, default=40_000_000)
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the CPU baseline (16 = one GPU's share of the box)")
    ap.add_argument("--host-threads", type=int, default=16, help="host threads of the synthetic generator")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    from minimap2_chaindp_amd import chaindp, params, shard

    if not torch.cuda.is_available() or chaindp.device_count() <= 0:
        raise SystemExit("bench.py needs a GPU: the chaining DP has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= n_dev:
        raise SystemExit(f"LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible")
    dev_index = local_rank % n_dev                       # one GPU per rank; several ranks share one only in a gloo rehearsal
    torch.cuda.set_device(dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)

    gen_preset = args.preset
    dp_preset = args.preset if args.preset in params.PRESETS else "ava-ont"      # skew / dense / ties use ava-ont gaps
    par = params.preset(dp_preset)

    # ---- this rank's shard of the job, generated on the host, then made resident in HBM
    n_reads = args.reads_per_gpu
    t_gen = time.time()
    off, anchors = shard.generate_shard(gen_preset, rank, world, n_reads, SEED, threads=args.host_threads)
    t_gen = time.time() - t_gen
    total = int(off[-1])
    n_reads = len(off) - 1                                   # (the skewed job is dealt by anchor count: read counts differ per rank)
    dev = chaindp.Device(dev_index, max_anchors=total + 1, max_reads=n_reads + 1, ring=args.ring or None)
    t_up = time.time()
    dev.upload(off, anchors)
    t_up = time.time() - t_up

    def barrier():
        torch.cuda.synchronize()
        dev.sync()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        dev.run_full(par)
    barrier()
    dev.set_profiling(True)
    dev.kernel_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dev.run_full(par)
    dev.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kms = dev.kernel_ms(reset=True)
    dev.set_profiling(False)
    stats = dev.stats()

    # max over ranks of the timed region; sum of anchors
    elapsed_max, total_all = shard.reduce_job(elapsed, total, dist if world > 1 else None,
                                              device="cuda" if args.backend == "nccl" else None)
    if world > 1:
        dist.barrier()

    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:      # side measurements only in the single-GPU run
        extras = measure_extras(torch, chaindp, dev, par, off, anchors, total)
        if gen_preset == "ava-ont" and n_reads == READS_PER_GPU:
            extras["other_configs"] = other_configs(chaindp, params, shard, dev_index, args)
            extras["map_batch"] = measure_map_batch(chaindp, params, dev_index)

    if rank == 0:
        steps = max(args.steps, 1)
        value = total_all * steps / elapsed_max
        dp_ms = kms["chain_dp"][0] / max(kms["chain_dp"][1], 1)
        pre_ms = kms["prepass"][0] / max(kms["prepass"][1], 1)
        cmp_ms = kms["compact"][0] / max(kms["compact"][1], 1)
        achieved = ALGO_BYTES_PER_ANCHOR * total / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0
        out = {
            "metric": "chained anchors/sec (whole node)",
            "value": value,
            "unit": "anchors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": f"{gen_preset} synthetic anchors, {n_reads} reads x 10 kb per GPU "
                            f"(shard of BASELINE configs[3]: 100k reads over 8 GPUs), DP preset {dp_preset}",
                "reads_per_gpu": n_reads, "anchors_per_gpu": total, "anchors_total": total_all,
                "units_per_gpu": stats["units"], "singletons_per_gpu": stats["singletons"],
                "seed": SEED, "parallelism": f"reads sharded over {world} GPU(s), no collective",
                "step": "prepass + chain DP + compaction (new_seed[]), inputs and outputs resident in HBM",
            },
            "roofline": {
                "kernel": "k_chain_twin (+ k_chain_units for the units it hands over)", "bound": "hbm",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(total),
                "algorithmic_bytes_per_anchor": ALGO_BYTES_PER_ANCHOR, "anchors_per_launch": total,
                "avg_launch_ms": dp_ms,
                # what actually bounds the kernel: instruction issue
                "issue": measured_issue(total, dp_ms),
            },
            "kernel_ms": {"prepass": pre_ms, "chain_dp": dp_ms, "compact": cmp_ms},
            "host": {"generate_s": t_gen, "upload_s": t_up, "upload_GBps": total * 16 / t_up / 1e9 if t_up > 0 else None},
        }
        out.update(extras)
        if "pair_evals_per_anchor" in extras:
            out["pair_evals_per_s"] = value * extras["pair_evals_per_anchor"]
        if "device_copy_GBps" in extras:
            out["roofline"]["frac_of_measured_copy_bw"] = achieved / extras["device_copy_GBps"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(par, off, anchors, args.cpu_sample_anchors, args.cpu_threads)
            # no published number exists for this metric, so vs_baseline stays null; the measured
            # ratios to the host's own CPU rate are reported under their own names
            cb = out["cpu_baseline"]
            out["vs_cpu_all_cores"] = value / cb["best_value"] if cb["best_value"] > 0 else None
            out["vs_cpu_single_core"] = value / cb["single_core_value"] if cb["single_core_value"] > 0 else None
        print(json.dumps(out), flush=True)

    dev.close()
    if world > 1:
        dist.destroy_process_group()


def other_configs(chaindp, params, shard, dev_index, args):
    """The other single-GPU shapes BASELINE.json names, same step (prepass + chain DP + compaction, resident), a few steps each:
    configs[2] (map-ont vs a human-size reference, ~50 M anchors), configs[4] (skewed 1e2..1e5 anchors per read) and the WHOLE job
    of configs[3] (100,000 reads, ~0.6 G anchors, ~10 GB of anchors) on one GPU.  Reported beside the headline, never as `value`."""
    out = {}
    for name, gen, preset, reads in (("map_ont_50M", "map-ont", "map-ont", 9_400), ("skew_1e2_1e5", "skew", "ava-ont", 3_000),
                                     ("full_100k_reads_1gpu", "ava-ont", "ava-ont", 100_000),
                                     # dense repeats (scans of hundreds to tens of thousands of predecessors): a batch that is all
                                     # tail -- 200 units of 15-38 k anchors -- and ten times that
                                     ("dense_200_units", "dense", "ava-ont", 100), ("dense_2000_units", "dense", "ava-ont", 1_000)):
        try:
            par = params.preset(preset)
            off, a = shard.generate_shard(gen, 0, 1, reads, SEED, threads=args.host_threads)
            tot = int(off[-1])
            with chaindp.Device(dev_index, max_anchors=tot + 1, max_reads=reads + 1) as d:
                d.upload(off, a)
                for _ in range(2):
                    d.run_full(par)
                d.sync()
                d.set_profiling(True); d.kernel_ms(reset=True)
                n = 5
                t0 = time.perf_counter()
                for _ in range(n):
                    d.run_full(par)
                d.sync()
                dt = time.perf_counter() - t0
                k = d.kernel_ms(reset=True)
                st = d.stats()
                out[name] = {"generator": gen, "dp_preset": preset, "reads": reads, "anchors": tot, "units": st["units"], "steps": n,
                             "ms_per_step": dt / n * 1e3, "anchors_per_s": tot * n / dt,
                             "kernel_ms": {kk: k[kk][0] / max(k[kk][1], 1) for kk in ("prepass", "chain_dp", "compact")},
                             "handed_to_one_unit_per_wave_kernel": d.leftover_units(), "handed_to_dense_kernel": d.deep_units()}
                if gen == "dense":                                   # the same batch with every unit left to its one wave (k_chain_units)
                    d.set_deep_handover(False)
                    d.run_full(par); d.sync()
                    t0 = time.perf_counter()
                    for _ in range(2):
                        d.run_full(par)
                    d.sync()
                    out[name]["ms_per_step_one_wave_per_unit"] = (time.perf_counter() - t0) / 2 * 1e3
                    out[name]["speedup_from_dense_kernel"] = out[name]["ms_per_step_one_wave_per_unit"] / out[name]["ms_per_step"]
            del off, a
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": repr(e)}
    return out


def measure_map_batch(chaindp, params, dev_index, target_anchors=24_000_000):
    """The resident pipeline as ONE call (chaindp_map_batch): the reference's minimizers in, its hits (mm_reg1_t) out; seed
    collection over the index image, chain DP, compaction, backtracking and mm_gen_regs never leave HBM.  Input: the reference's
    own dump of an all-vs-all run (tests/golden/_big, 600 reads x 8 kb, when present on the box) or else the committed
    synthetic-repeat fixture, repeated to a batch of ~24 M anchors.  PCIe carries ~5 B in and ~1 B out per anchor instead of
    16 in and 24 out."""
    big = os.path.join(ROOT, "tests", "golden", "_big", "big_avaont.npz")
    path = big if os.path.exists(big) else os.path.join(ROOT, "tests", "golden", "seeds", "syn_repeats_avaont.npz")
    try:
        g = np.load(path, allow_pickle=False)
        pv = [int(x) for x in g["params"]]
        par = params.ChainParams(max_dist_x=pv[0], max_dist_y=pv[1], bw=pv[2], max_skip=pv[3], min_sc=pv[4], is_cdna=pv[5], n_segs=1)
        mult = max(1, int(target_anchors // max(len(g["anchors"]), 1)))
        mini_off = np.concatenate([[0], np.cumsum(np.tile(np.diff(g["mini_off"]), mult))]).astype(np.int64)
        mini, bid, qlen = np.tile(g["mini"], (mult, 1)), np.tile(g["bid"], mult), np.tile(g["qlen"], mult)
        n_reads = len(bid)
        hash_ = (np.arange(n_reads, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(1 << 32)).astype(np.uint32)
        cap_a = len(g["anchors"]) * mult + 1024
        with chaindp.Device(dev_index, max_anchors=cap_a, max_reads=n_reads + 1) as d:
            ix = d.load_index([g["img_B"], g["img_H"], g["img_V"], g["img_P"]])
            roff, regs, rep, na = d.map_batch(ix, int(g["flag"]), int(g["mid_occ"]), par, pv[7], mini_off, mini, bid, qlen, hash_, regs_cap=cap_a // 8)
            n = 3
            t0 = time.perf_counter()
            for _ in range(n):
                roff, regs, rep, na = d.map_batch(ix, int(g["flag"]), int(g["mid_occ"]), par, pv[7], mini_off, mini, bid, qlen, hash_, regs_cap=cap_a // 8)
            dt = (time.perf_counter() - t0) / n
            # ... and the same call from three host threads, a context each, the index image shared: while one context waits for its
            # seed counts or downloads its hits, the other two have kernels on the GPU (what the packet shim's service contexts do)
            conc = None
            try:
                import threading
                n_ctx, n_each = 3, 4
                devs = [chaindp.Device(dev_index, max_anchors=cap_a, max_reads=n_reads + 1) for _ in range(n_ctx)]
                try:
                    def work(dd, k):
                        for _ in range(k):
                            dd.map_batch(ix, int(g["flag"]), int(g["mid_occ"]), par, pv[7], mini_off, mini, bid, qlen, hash_, regs_cap=cap_a // 8)
                    for dd in devs:
                        work(dd, 1)                                   # first use: allocations
                    th = [threading.Thread(target=work, args=(dd, n_each)) for dd in devs]
                    t0 = time.perf_counter()
                    for t in th:
                        t.start()
                    for t in th:
                        t.join()
                    dtc = time.perf_counter() - t0
                    conc = {"contexts": n_ctx, "batches": n_ctx * n_each, "seconds": dtc, "anchors_per_s": na * n_ctx * n_each / dtc}
                finally:
                    for dd in devs:
                        dd.close()
            except Exception as e:  # noqa: BLE001
                conc = {"error": repr(e)}
        return {"three_contexts": conc, "input": os.path.relpath(path, ROOT), "repeated": mult, "reads": n_reads, "minimizers": int(mini_off[-1]), "anchors": na,
                "hits": int(roff[-1]), "seconds_per_batch": dt, "anchors_per_s": na / dt, "minimizers_per_s": int(mini_off[-1]) / dt,
                "bytes_in_per_anchor": (16 * int(mini_off[-1]) + 20 * n_reads) / max(na, 1), "bytes_out_per_anchor": (80 * int(roff[-1]) + 12 * n_reads) / max(na, 1),
                "includes": "H2D of minimizers (pageable), collect_seed_hits + sort, prepass + chain DP + compaction, backtrack, mm_gen_regs, D2H of hits"}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def measure_extras(torch, chaindp, dev, par, off, anchors, total):
    """Side measurements BASELINE.md asks to report next to the headline (none of them is `value`):
    the PCIe-inclusive end-to-end rate (host anchors in, f/p/v and new_seed[] back out, pageable host memory),
    the device copy bandwidth (practical HBM ceiling), and pair evaluations (executions of chain.c:254)."""
    ex = {}
    t0 = time.perf_counter()
    f, p, v = dev.chain_batch(par, off, anchors)
    soff, seeds = dev.compact(par)
    dt = time.perf_counter() - t0
    ex["end_to_end"] = {"anchors_per_s": total / dt, "seconds": dt,
                        "includes": "H2D of anchors, prepass + chain DP + compaction, D2H of f/p/v and new_seed[] (pageable host buffers)"}
    try:
        ex["end_to_end_pipelined"] = measure_pipelined(torch, chaindp, dev.device, par, off, anchors, total, ex["end_to_end"]["anchors_per_s"])
    except Exception as e:  # noqa: BLE001
        ex["end_to_end_pipelined"] = {"error": repr(e)}
    # the host half (mm_chain_dp_bottom, chain.c:329-431) on the GPU as well, timed on its own
    try:
        dev.set_profiling(True); dev.kernel_ms(reset=True)
        t0 = time.perf_counter()
        coff, u, boff, b = dev.backtrack(par, 3)
        dt_bt = time.perf_counter() - t0
        kb = dev.kernel_ms(reset=True)["backtrack"]
        dev.set_profiling(False)
        ex["backtrack"] = {"kernels_ms": kb[0] / max(kb[1], 1), "with_download_s": dt_bt, "chains": int(coff[-1]),
                           "chained_anchors": int(boff[-1]), "anchors_per_s_kernels": total / (kb[0] / max(kb[1], 1) * 1e-3)}
        # ... and the chains as hits (mm_gen_regs, hit.c:52-95), with the 80 B/hit download
        n_r = len(off) - 1
        t0 = time.perf_counter()
        regs = dev.gen_regs(np.arange(n_r, dtype=np.uint32), np.full(n_r, 1 << 20, np.int32), coff[-1])
        ex["backtrack"]["gen_regs_with_download_s"] = time.perf_counter() - t0
        ex["backtrack"]["hits"] = int(len(regs))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol
        k = min(400, len(soff) - 1)
        t0 = time.perf_counter()
        for r in range(k):
            ol.oracle_bottom(3, par.min_sc, seeds[int(soff[r]):int(soff[r + 1])])
        ex["backtrack"]["cpu_port_1core_anchors_per_s"] = int(off[k]) / (time.perf_counter() - t0)
    except Exception as e:  # noqa: BLE001
        ex["backtrack"] = {"error": str(e)}
    n = 1 << 28                                                  # 1 GiB of int32 each way
    src = torch.empty(n, dtype=torch.int32, device="cuda")
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    ex["device_copy_GBps"] = 5 * 2 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9      # read + write bytes
    del src, dst
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol
        k = min(200, len(off) - 1)
        so = np.ascontiguousarray(off[:k + 1])
        _, _, _, evals = ol.oracle_batch(par, so, np.ascontiguousarray(anchors[:int(so[-1])]), threads=8)
        ex["pair_evals_per_anchor"] = evals / max(int(so[-1]), 1)     # inner-loop executions of the scalar algorithm
    except Exception as e:  # noqa: BLE001
        ex["pair_evals_error"] = repr(e)
    return ex


def pinned_copy_bandwidth(torch, nbytes=1 << 30, reps=3):
    """GB/s of hipMemcpyAsync between pinned host memory and HBM, each direction alone (the PCIe ceiling of this box)."""
    h = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    out = {}
    for name, src, dst in (("h2d", h, d), ("d2h", d, h)):
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        out[name] = reps * nbytes / (time.perf_counter() - t0) / 1e9
    del h, d
    return out


def measure_pipelined(torch, chaindp, dev_index, par, off, anchors, total, pageable_rate, n_batches=12, depth=3):
    """SURVEY 8d's first metric: anchors/s of the device stage INCLUDING transfers, the way a driver would run it
    (reference fpga_chaindp.c:102-159 / 228-266): batches of the bench size stream through chaindp_pipe_t (three
    contexts, three streams), anchors come from pinned host memory, new_seed[] records land in pinned host memory.
    Never `value`."""
    bw = pinned_copy_bandwidth(torch)
    n_reads = len(off) - 1
    pin_off = chaindp.PinnedArray((n_reads + 1,), np.int64)
    pin_a = chaindp.PinnedArray((total, 2), np.uint64)
    pin_off.array[:] = off
    pin_a.array[:] = np.ascontiguousarray(anchors, np.uint64).reshape(-1, 2)
    res = {}
    with chaindp.Pipe(dev_index, depth=depth, max_anchors=total + 1, max_reads=n_reads + 1) as pipe:
        def stream(nb):
            sub = done = 0
            seeds_total = 0
            while done < nb:
                while sub < nb and pipe.submit(par, pin_off.array, pin_a.array, tag=sub):
                    sub += 1
                _, soff, _ = pipe.wait(copy=False)
                seeds_total += int(soff[-1])
                pipe.release()
                done += 1
            return seeds_total
        stream(depth)                                            # warm-up: first-use allocations, page faults of the pinned results
        t0 = time.perf_counter()
        seeds_total = stream(n_batches)
        dt = time.perf_counter() - t0
    bytes_in, bytes_out = n_batches * (total * 16 + (n_reads + 1) * 8), seeds_total * 24 + n_batches * (n_reads + 1) * 8
    t_pcie = max(bytes_in / (bw["h2d"] * 1e9), bytes_out / (bw["d2h"] * 1e9))      # full duplex: the slower direction bounds it
    res = {"anchors_per_s": n_batches * total / dt, "seconds": dt, "batches": n_batches, "depth": depth,
           "anchors_per_batch": total, "seeds_per_batch": seeds_total // n_batches,
           "bytes_in_per_anchor": 16, "bytes_out_per_anchor": 24 * seeds_total / (n_batches * total),
           "pinned_h2d_GBps": bw["h2d"], "pinned_d2h_GBps": bw["d2h"],
           "achieved_h2d_GBps": bytes_in / dt / 1e9, "achieved_d2h_GBps": bytes_out / dt / 1e9,
           "pcie_frac": t_pcie / dt, "vs_pageable_sync": n_batches * total / dt / pageable_rate if pageable_rate else None,
           "includes": "H2D of anchors from pinned memory, prepass + chain DP + compaction, D2H of new_seed[] into pinned memory; "
                       "three contexts / streams, upload(n+1) | kernels(n) | download(n-1) overlapped"}
    pin_off.free(); pin_a.free()
    return res


def kernel_source_sha16():
    import hashlib
    h = hashlib.sha256()
    for f in ("chaindp_twin.hip", "chaindp_kernels.hip", "chaindp_fast.h", "chaindp_wave.h"):
        h.update(open(os.path.join(ROOT, "minimap2_chaindp_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def stored_pmc(anchors_per_launch):
    """PMC figures of the DP kernel from tools/profile.sh on this same workload (profiles/latest_traffic.json).  PMC collection
    cannot run inside the timed process, so they are quoted only if they were measured on THIS build of the kernels (hash of
    their sources) and this batch size; otherwise None."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "latest_traffic.json")))
        if t.get("anchors_per_launch") == anchors_per_launch and t.get("kernel_source_sha16") == kernel_source_sha16():
            return t
    except Exception:  # noqa: BLE001
        pass
    return None


def measured_traffic(anchors_per_launch):
    """HBM bytes per launch of the DP kernel (FETCH_SIZE/WRITE_SIZE collected and corrected as MI355X_MICROARCH.md prescribes)."""
    t = stored_pmc(anchors_per_launch)
    return t["hbm_bytes_per_launch"] if t else None


# instruction-issue ceilings of the chip, measured by tools/issue_calib.hip (profiles/r02_issue_calib_*.json), G wave-instructions/s
VALU_FULL_RATE, VALU_HALF_RATE, SALU_RATE = 962.0, 590.0, 574.0


def measured_issue(anchors_per_launch, dp_ms):
    """What actually bounds the DP kernel: instruction issue, not HBM.  Instruction counts from the stored PMC pass (same build,
    same batch), duration live.  A VALU instruction issues at the full rate only if it is a plain two-source one; SGPR operands,
    three sources, DPP and compares run at half of it, so the VALU ceiling lies between the two figures; the scalar unit is
    shared by a CU's four SIMDs."""
    t = stored_pmc(anchors_per_launch)
    if not t or dp_ms <= 0 or not t.get("valu_insts_per_launch"):
        return None
    sec = dp_ms * 1e-3
    out = {"valu": {"insts_per_anchor": t["valu_insts_per_launch"] / anchors_per_launch, "achieved": t["valu_insts_per_launch"] / sec / 1e9,
                    "ceiling_all_full_rate": VALU_FULL_RATE, "ceiling_all_half_rate": VALU_HALF_RATE},
           "unit": "G wave-instructions/s", "ceilings_from": "tools/issue_calib.hip on MI355X (profiles/r02_issue_calib_*.json)"}
    if t.get("salu_insts_per_launch"):
        out["salu"] = {"insts_per_anchor": t["salu_insts_per_launch"] / anchors_per_launch, "achieved": t["salu_insts_per_launch"] / sec / 1e9,
                       "ceiling": SALU_RATE, "frac": t["salu_insts_per_launch"] / sec / 1e9 / SALU_RATE}
    return out


def cpu_baseline(par, off, anchors, sample_anchors, threads):
    """The reference's per-read call (malloc, recurrence, compaction, free; chain.c:218-327) over the first
    reads of the same batch that hold about `sample_anchors` anchors, timed three times: on one thread, on
    `threads` threads (default 16: one GPU's share of an 8-GPU box) and on every logical CPU the process may
    use (the authors ran -t 56, run.sh:3) -- each leg sized to roughly 10-20 s of CPU work.  `value` is the
    ALL-CORE figure: that is what the GPU has to beat as a system."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    n = int(np.searchsorted(off, sample_anchors, side="left"))
    n = max(1, min(n, len(off) - 1))
    soff = np.ascontiguousarray(off[:n + 1])
    sa = np.ascontiguousarray(anchors[:int(soff[-1])])
    use_ref = ol.have_ref()
    n1 = max(1, n // 64)                                        # single-core probe on 1/64 of the sample
    sec1, _ = ol.time_top(par, np.ascontiguousarray(soff[:n1 + 1]), sa, threads=1, use_ref=use_ref)
    rate1 = int(soff[n1]) / sec1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    legs = {}
    for th in sorted({max(1, threads), usable} | {t for t in (32, 64, 128) if t < usable}):
        th = min(th, n)                                         # at least one read per thread
        reps = int(max(1, min(8, round(15.0 * rate1 / int(soff[-1])))))   # ~15 core-seconds per leg
        sec, _ = ol.time_top(par, soff, sa, threads=th, use_ref=use_ref, reps=reps)
        legs[th] = {"threads": th, "anchors_per_s": int(soff[-1]) * reps / sec, "seconds": sec, "passes": reps,
                    "cpu_core_seconds": sec * th}
    best_th = max(legs, key=lambda t: legs[t]["anchors_per_s"])
    all_th = max(legs)
    return {
        "value": legs[all_th]["anchors_per_s"], "unit": "anchors/s", "cores": all_th,
        "kind": "reference" if use_ref else "port",
        "sample": f"first {n} reads ({int(soff[-1])} anchors) of the same batch x {legs[all_th]['passes']} passes, per-read "
                  f"mm_chain_dp_fpga call (malloc + recurrence + compaction + free), reads dealt to {all_th} threads "
                  f"by anchor count, clock from all-workers-ready to last-worker-done",
        "seconds": legs[all_th]["seconds"], "cpu_core_seconds": legs[all_th]["cpu_core_seconds"],
        "single_core_value": rate1, "by_threads": [legs[t] for t in sorted(legs)],
        "best_threads": best_th, "best_value": legs[best_th]["anchors_per_s"],
        "host_logical_cpus": os.cpu_count(), "usable_cpus": usable,
        "note": "value is the rate with every logical CPU the process may use; where more threads are SLOWER than fewer (by_threads), the "
                "box's CPU time is capped for this job (a one-GPU lease gets a share of the host) and best_value is the CPU rate to compare with",
    }


if __name__ == "__main__":
    main()
