#!/usr/bin/env python3
"""bench.py -- chained anchors/s of the device half of minimap2's chaining (reference chain.c:218-327)
on N MI355X, one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[3], "ava-ont self-overlap, 100k synthetic 10 kb reads, read-sharded
across 8 GPUs"): ONE fixed job of 100,000 reads (~6.1k anchors per read, 0.61 G anchors, 9.8 GB of
anchors), cut over the N ranks by cumulative anchor count (--scaling strong, the default: at N=1 the
whole job sits on one GPU, at N=8 each rank holds ~12,500 reads).  --scaling weak gives every rank
its own 12,500 reads instead (per-GPU work fixed).  A rank generates its shard on the host with the
job's seed and uploads it to HBM BEFORE the clock starts.  A step = one pass of the whole device half
over the resident shard: prepass (unit split, avg_qspan sums) + chain DP (f/p/v) + compaction into
new_seed[], all HIP kernels, results left in HBM.  Reads are independent, so there is no collective
on the data path; torch.distributed (RCCL) only provides the barrier and the max-over-ranks time.

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

READS_PER_GPU = 12_500          # 100k reads / 8 GPUs (configs[3]): the weak-scaling shard, and the sub-batch of the side measurements
JOB_READS = {"ava-ont": 100_000, "map-ont": 9_400, "skew": 24_000, "dense": 12_500}   # the fixed job of --scaling strong, by preset
SEED = 20261004
ALGO_BYTES_PER_ANCHOR = 24
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong: one fixed job (--job-reads) cut over the ranks by anchor count; weak: --reads-per-gpu reads on every rank")
    ap.add_argument("--job-reads", type=int, default=0, help="reads of the fixed job (strong scaling); default by preset: 100000 for ava-ont (BASELINE configs[3])")
    ap.add_argument("--reads-per-gpu", type=int, default=READS_PER_GPU, help="reads per rank (weak scaling)")
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
    strong = args.scaling == "strong"
    job_reads = (args.job_reads or JOB_READS.get(gen_preset, 100_000)) if strong else args.reads_per_gpu * world
    t_gen = time.time()
    if strong:
        off, anchors, _first = shard.generate_job_shard(gen_preset, rank, world, job_reads, SEED, threads=args.host_threads)
    else:
        off, anchors = shard.generate_shard(gen_preset, rank, world, args.reads_per_gpu, SEED, threads=args.host_threads)
    t_gen = time.time() - t_gen
    total = int(off[-1])
    n_reads = len(off) - 1                                   # (a job dealt by anchor count: read counts differ per rank)
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
    handed_over = {"to_one_unit_per_wave_kernel": dev.leftover_units(), "to_dense_kernel": dev.deep_units()}   # this rank's last timed step

    # max over ranks of the timed region; sum of anchors
    elapsed_max, total_all = shard.reduce_job(elapsed, total, dist if world > 1 else None,
                                              device="cuda" if args.backend == "nccl" else None)
    if world > 1:
        dist.barrier()

    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:      # side measurements only in the single-GPU run, on one GPU's share of the job
        ns = min(n_reads, READS_PER_GPU)
        sub_off = np.ascontiguousarray(off[:ns + 1])
        sub_a = anchors[:int(sub_off[-1])]
        extras = measure_extras(torch, chaindp, dev, par, sub_off, sub_a, int(sub_off[-1]))
        extras["side_measurements_on"] = f"the first {ns} reads of the job ({int(sub_off[-1])} anchors): one GPU's share at N=8"
        if gen_preset == "ava-ont" and strong and job_reads == JOB_READS["ava-ont"]:
            extras["other_configs"] = other_configs(chaindp, params, shard, dev_index, args)
            extras["map_batch"] = measure_map_batch(chaindp, params, dev_index)
            extras["packet_abi"] = measure_packet_abi(par, sub_off, sub_a)

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
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": (f"{gen_preset} synthetic anchors, ONE job of {job_reads} reads x 10 kb"
                             + (" (BASELINE configs[3])" if gen_preset == "ava-ont" and job_reads == 100_000 else "")
                             + f" cut over {world} GPU(s) by anchor count, DP preset {dp_preset}") if strong else
                            (f"{gen_preset} synthetic anchors, {n_reads} reads x 10 kb per GPU "
                             f"(weak scaling: every rank its own shard; at 8 GPUs the 100k reads of BASELINE configs[3]), DP preset {dp_preset}"),
                "job_reads": job_reads, "reads_on_rank0": n_reads, "anchors_on_rank0": total, "anchors_total": total_all,
                "units_on_rank0": stats["units"], "singletons_on_rank0": stats["singletons"],
                "seed": SEED, "parallelism": f"reads sharded over {world} GPU(s), no collective",
                "step": "prepass + chain DP + compaction (new_seed[]), inputs and outputs resident in HBM",
            },
            "roofline": {
                "kernel": "k_chain_twin (+ k_chain_units for the units it hands over)", "bound": "hbm",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(total),
                "algorithmic_bytes_per_anchor": ALGO_BYTES_PER_ANCHOR, "anchors_per_launch": total,
                "avg_launch_ms": dp_ms,
                # "bound" names the roofline the contract asks for (HBM); what actually limits the kernel is
                # instruction issue, reported beside it
                "limited_by": "VALU instruction issue (see issue)", "issue": measured_issue(total, dp_ms),
            },
            "kernel_ms": {"prepass": pre_ms, "chain_dp": dp_ms, "compact": cmp_ms},
            "handed_over": handed_over,
            "host": {"generate_s": t_gen, "upload_s": t_up, "upload_GBps": total * 16 / t_up / 1e9 if t_up > 0 else None},
        }
        out.update(extras)
        if "pair_evals_per_anchor" in extras:
            out["pair_evals_per_s"] = value * extras["pair_evals_per_anchor"]
        st_pmc = stored_pmc_any()
        sp = stored_pmc(total)
        if sp is not None and sp.get("scaled_from_anchors"):
            out["roofline"]["traffic_scaled_from_anchors"] = sp["scaled_from_anchors"]      # per-anchor counters of the same build's N=1 profile
        if out["roofline"]["traffic"] is None and st_pmc is not None:
            # the stored counters belong to another build of the kernels or another batch: say so instead of a silent null
            out["roofline"]["traffic_stale"] = True
            out["roofline"]["traffic_last_measured"] = {k: st_pmc.get(k) for k in ("hbm_bytes_per_launch", "anchors_per_launch", "kernel_source_sha16")}
            out["roofline"]["kernel_source_sha16_now"] = kernel_source_sha16()
        if "device_copy_GBps" in extras:
            out["roofline"]["frac_of_measured_copy_bw"] = achieved / extras["device_copy_GBps"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(par, off, anchors, args.cpu_sample_anchors, args.cpu_threads)
            # no published number exists for this metric, so vs_baseline stays null; the measured
            # ratios to the host's own CPU rate are reported under their own names
            cb = out["cpu_baseline"]
            out["vs_cpu_best"] = value / cb["best_value"] if cb["best_value"] > 0 else None          # against the best thread count's rate
            out["vs_cpu_all_cores"] = value / cb["value"] if cb["value"] > 0 else None             # against the all-logical-CPUs leg
            out["vs_cpu_single_core"] = value / cb["single_core_value"] if cb["single_core_value"] > 0 else None
        print(json.dumps(out), flush=True)

    dev.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_reference_bounded(par, off, a, threads, budget_s=2.5):
    """The reference's mm_chain_dp_fpga (oracle/_ref; the oracle's port where that build is absent) on `threads` host threads over a
    bounded sample of a batch: a short probe sizes the sample to about budget_s seconds of wall time.  A reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    use_ref = ol.have_ref()
    n_all = len(off) - 1
    tot = int(off[-1])
    n1 = int(np.searchsorted(off, max(tot // 64, 1), side="left"))
    n1 = max(min(threads, n_all), min(n1, n_all))
    so = np.ascontiguousarray(off[:n1 + 1])
    th = max(1, min(threads, n1))
    sec, _ = ol.time_top(par, so, a, threads=th, use_ref=use_ref)
    rate = int(so[-1]) / max(sec, 1e-9)
    n2 = int(np.searchsorted(off, min(tot, max(int(rate * budget_s), 1)), side="left"))
    n2 = max(n1, min(n2, n_all))
    so = np.ascontiguousarray(off[:n2 + 1])
    th = max(1, min(threads, n2))
    sec, _ = ol.time_top(par, so, a, threads=th, use_ref=use_ref)
    return {"anchors_per_s": int(so[-1]) / max(sec, 1e-9), "threads": th, "kind": "reference" if use_ref else "port",
            "sample": f"first {n2} reads ({int(so[-1])} anchors), one pass, per-read mm_chain_dp_fpga call", "seconds": sec}


def other_configs(chaindp, params, shard, dev_index, args):
    """The other single-GPU shapes BASELINE.json names, same step (prepass + chain DP + compaction, resident), a few steps each:
    configs[2] (map-ont vs a human-size reference, ~50 M anchors), configs[4] (skewed 1e2..1e5 anchors per read), one GPU's share
    (12,500 reads) of the configs[3] job -- round 1 and 2's headline workload --, and dense repeats.  Each with the reference's CPU
    rate on the same batch beside it (bounded sample, --cpu-threads threads).  Reported beside the headline, never as `value`."""
    out = {}
    for name, gen, preset, reads in (("map_ont_50M", "map-ont", "map-ont", 9_400), ("skew_1e2_1e5", "skew", "ava-ont", 3_000),
                                     ("shard_12500_reads_1gpu", "ava-ont", "ava-ont", READS_PER_GPU),
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
            if not args.no_cpu_baseline:
                try:
                    out[name]["cpu_reference"] = cpu_reference_bounded(par, off, a, args.cpu_threads)
                    out[name]["vs_cpu_reference"] = out[name]["anchors_per_s"] / out[name]["cpu_reference"]["anchors_per_s"]
                except Exception as e:  # noqa: BLE001
                    out[name]["cpu_reference"] = {"error": repr(e)}
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


def _replay_file(path, packets, blobs, flag, mid_occ, par):
    import struct
    with open(path, "wb") as fh:
        fh.write(b"SHIMRPL1" + struct.pack("<6i", int(flag), int(mid_occ), par.bw, par.max_skip, par.min_sc, len(packets)))
        for blob in blobs:
            blob = np.ascontiguousarray(blob, np.uint8)
            fh.write(struct.pack("<q", blob.size))
            fh.write(blob.tobytes())
        for pk in packets:
            fh.write(struct.pack("<I", len(pk)))
            fh.write(pk)


def _run_replay(path, producers, reps, rounds, anchors_per_rep, services=2, max_packets=256):
    import subprocess
    exe = os.path.join(ROOT, "minimap2_chaindp_amd", "csrc", "shim_replay")
    r = subprocess.run([exe, path, str(producers), str(reps), str(rounds), str(max_packets), "0", str(services)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    if r.returncode != 0:
        return {"error": f"shim_replay exited with {r.returncode}: {r.stderr[-300:]}"}
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    rounds_ = [ln for ln in lines if "round" in ln]
    shim = next((ln["shim"] for ln in lines if "shim" in ln), None)
    steady = rounds_[len(rounds_) // 2:] or rounds_                # the first rounds pay the first-use allocations (pinned pool, scratch, index upload)
    best = min(steady, key=lambda d: d["seconds"])
    secs = sorted(d["seconds"] for d in steady)
    med = secs[len(secs) // 2]
    return {"anchors_per_s": anchors_per_rep * reps / med, "best_round_anchors_per_s": anchors_per_rep * reps / best["seconds"],
            "records_out_per_s": best["records_out"] / best["seconds"], "elements_in_per_s": best["elements_in"] / best["seconds"],
            "anchors_per_round": anchors_per_rep * reps, "packets_per_round": best["packets"], "reads_per_packet": 8, "producers": producers,
            "service_contexts": services, "max_packets_per_device_batch": max_packets,
            "rounds": [round(d["seconds"], 5) for d in rounds_], "PCIe_GBps_in": best["bytes_in"] / best["seconds"] / 1e9,
            "PCIe_GBps_out": best["records_out"] * 24 / best["seconds"] / 1e9, "err_reads": best["err_reads_so_far"],
            "device_batches": shim["device_batches"] if shim else None, "per_gpu": shim["gpus"] if shim else None}


def measure_packet_abi(par, off, anchors, target_anchors=20_000_000, producers=8):
    """The product's real boundary: the reference's driver ABI (fpga.h:37-62).  A C host (tools/shim_replay.c, built by the csrc
    Makefile) plays the reference's threads -- `producers` producer threads that get a driver buffer, memcpy a packet of 8 reads
    into it and submit it (map.c:423-444), one receiver that walks and releases every result packet (fpga_chaindp.c:228-266) --
    against libchaindp_hip.so in a process of its own (two service contexts per GPU for anchor packets, three for minimizer packets:
    chaindp_fpga_configure_services).  Two packet kinds: anchor packets (type 0x41; the first reads of the bench
    job, >= 20 M anchors per pass) and the reference's own minimizer packets (type 3) with the index image streamed through
    fpga_load_index first (the reference's dump of an all-vs-all run when it is on the box, else the committed synthetic-repeat
    fixture).  PCIe-inclusive steady-state rates (median round after the first); never `value`."""
    import tempfile
    from minimap2_chaindp_amd import fpga
    out = {}
    exe = os.path.join(ROOT, "minimap2_chaindp_amd", "csrc", "shim_replay")
    if not os.path.exists(exe):
        return {"error": "minimap2_chaindp_amd/csrc/shim_replay is not built (make -C minimap2_chaindp_amd/csrc)"}
    tmpdir = tempfile.mkdtemp(prefix="chaindp_replay_")
    try:
        # ---- anchor packets
        try:
            n = int(np.searchsorted(off, target_anchors, side="left"))
            n = max(8, min(n, len(off) - 1))
            packets = [fpga.build_task_packet([(r, anchors[int(off[r]):int(off[r + 1])]) for r in range(k, min(k + 8, n))],
                                              par.max_dist_x, par.max_dist_y) for k in range(0, n, 8)]
            path = os.path.join(tmpdir, "anchors.rpl")
            _replay_file(path, packets, [np.zeros(0, np.uint8)] * 4, 0, 0, par)
            del packets
            out["anchor_packets"] = _run_replay(path, producers, 25, 5, int(off[n]), services=2)
            out["anchor_packets"]["bytes_in_per_anchor"], out["anchor_packets"]["input"] = 16, f"first {n} reads of the bench job"
            os.unlink(path)
        except Exception as e:  # noqa: BLE001
            out["anchor_packets"] = {"error": repr(e)}
        # ---- the reference's minimizer packets (type 3)
        try:
            big = os.path.join(ROOT, "tests", "golden", "_big", "big_avaont.npz")
            src = big if os.path.exists(big) else os.path.join(ROOT, "tests", "golden", "seeds", "syn_repeats_avaont.npz")
            g = np.load(src, allow_pickle=False)
            pv = [int(x) for x in g["params"]]
            from minimap2_chaindp_amd import params as P
            mpar = P.ChainParams(max_dist_x=pv[0], max_dist_y=pv[1], bw=pv[2], max_skip=pv[3], min_sc=pv[4], is_cdna=pv[5], n_segs=1)
            nr = len(g["bid"])
            reads = [(r, g["mini"][g["mini_off"][r]:g["mini_off"][r + 1]], int(g["bid"][r]), int(g["qlen"][r])) for r in range(nr)]
            packets = [fpga.build_task_packet(reads[k:k + 8], mpar.max_dist_x, mpar.max_dist_y, pkt_type=fpga.PKT_MINIMIZERS) for k in range(0, nr, 8)]
            tot_a = int(g["a_off"][-1]) if "a_off" in g.files else len(g["anchors"])
            reps = max(1, int(round(25 * target_anchors / max(tot_a, 1))))       # ~500 M anchors a round: filling and draining the contexts' pipeline
                                                                                 # (a device batch is ~10 M anchors, three in flight) must not weigh
            path = os.path.join(tmpdir, "minimizers.rpl")
            _replay_file(path, packets, [g["img_B"], g["img_H"], g["img_V"], g["img_P"]], g["flag"], g["mid_occ"], mpar)
            out["minimizer_packets"] = _run_replay(path, producers, reps, 5, tot_a, services=3)
            out["minimizer_packets"]["minimizers_per_s"] = out["minimizer_packets"].get("elements_in_per_s")
            out["minimizer_packets"]["input"] = f"{os.path.relpath(src, ROOT)} ({nr} reads, {int(g['mini_off'][-1])} minimizers -> {tot_a} anchors) x {reps} per round"
            os.unlink(path)
        except Exception as e:  # noqa: BLE001
            out["minimizer_packets"] = {"error": repr(e)}
    finally:
        try:
            os.rmdir(tmpdir)
        except OSError:
            pass
    out["includes"] = ("producer memcpy into driver buffers, packet parsing, H2D by a gather kernel over the pinned packets, (seed collection,) prepass + "
                       "chain DP + compaction, result packets assembled in pinned memory by a scatter kernel, receiver walk and release")
    return out


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
        dev.backtrack(par, 3)                                        # first use: allocations (they would sit inside the event window)
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
    for f in ("chaindp_twin.hip", "chaindp_kernels.hip", "chaindp_fast.h", "chaindp_wave.h", "chaindp_lanes.h"):
        h.update(open(os.path.join(ROOT, "minimap2_chaindp_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def stored_pmc_any():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "latest_traffic.json")))
    except Exception:  # noqa: BLE001
        return None


def stored_pmc(anchors_per_launch):
    """PMC figures of the DP kernel from tools/profile.sh (profiles/latest_traffic.json).  PMC collection cannot run inside the timed
    process, so they are quoted only if they were measured on THIS build of the kernels (hash of their sources); for another batch
    size of the same workload (a rank's share of the job at N > 1) the per-anchor figures of that profile are scaled to the batch,
    and the record says so (`scaled_from_anchors`).  Otherwise None."""
    t = stored_pmc_any()
    try:
        if not t or t.get("kernel_source_sha16") != kernel_source_sha16():
            return None
        n0 = t.get("anchors_per_launch")
        if n0 == anchors_per_launch:
            return t
        if n0 and anchors_per_launch > 0:
            k = anchors_per_launch / n0
            u = dict(t)
            for key in ("hbm_read_bytes", "hbm_write_bytes", "hbm_bytes_per_launch", "valu_insts_per_launch", "salu_insts_per_launch", "lds_insts_per_launch"):
                if u.get(key) is not None:
                    u[key] = u[key] * k
            u["anchors_per_launch"] = anchors_per_launch
            u["scaled_from_anchors"] = n0
            return u
    except Exception:  # noqa: BLE001
        pass
    return None


def measured_traffic(anchors_per_launch):
    """HBM bytes per launch of the DP kernel (FETCH_SIZE/WRITE_SIZE collected and corrected as MI355X_MICROARCH.md prescribes)."""
    t = stored_pmc(anchors_per_launch)
    return t["hbm_bytes_per_launch"] if t else None


def issue_ceilings():
    """Instruction-issue ceilings of the chip in G wave-instructions/s, from the calibration runs of tools/issue_calib.hip that are
    committed under profiles/ (8 waves per SIMD, every CU busy): plain two-source VALU (v_add_u32), the slow VALU class (v_min3_i32:
    three sources; SGPR operands, DPP and compares issue alike) and SALU (s_add_u32); mean over the runs."""
    import glob
    acc = {"v_add_u32": [], "v_min3_i32": [], "s_add_u32": []}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02_issue_calib_*.json")))
    for f in files:
        try:
            for r in json.load(open(f))["results"]:
                if r.get("waves_per_simd") == 8 and r.get("kind") in acc:
                    acc[r["kind"]].append(r["salu_Ginst_s"] if r["kind"].startswith("s_") else r["valu_Ginst_s"])
        except Exception:  # noqa: BLE001
            pass
    if not all(acc.values()):
        return None
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    return {"valu_full": m["v_add_u32"], "valu_half": m["v_min3_i32"], "salu": m["s_add_u32"], "files": [os.path.relpath(f, ROOT) for f in files]}



def measured_issue(anchors_per_launch, dp_ms):
    """What actually bounds the DP kernel: instruction issue, not HBM.  Instruction counts from the stored PMC pass (same build,
    same batch), duration live.  A VALU instruction issues at the full rate only if it is a plain two-source one; SGPR operands,
    three sources, DPP and compares run at half of it, so the VALU ceiling lies between the two figures; the scalar unit is
    shared by a CU's four SIMDs."""
    t = stored_pmc(anchors_per_launch)
    ceil = issue_ceilings()
    if not t or not ceil or dp_ms <= 0 or not t.get("valu_insts_per_launch"):
        return None
    sec = dp_ms * 1e-3
    out = {"valu": {"insts_per_anchor": t["valu_insts_per_launch"] / anchors_per_launch, "achieved": t["valu_insts_per_launch"] / sec / 1e9,
                    "ceiling_all_full_rate": ceil["valu_full"], "ceiling_all_half_rate": ceil["valu_half"]},
           "unit": "G wave-instructions/s", "ceilings_from": "tools/issue_calib.hip on MI355X: " + ", ".join(ceil["files"])}
    if t.get("salu_insts_per_launch"):
        out["salu"] = {"insts_per_anchor": t["salu_insts_per_launch"] / anchors_per_launch, "achieved": t["salu_insts_per_launch"] / sec / 1e9,
                       "ceiling": ceil["salu"], "frac": t["salu_insts_per_launch"] / sec / 1e9 / ceil["salu"]}
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
