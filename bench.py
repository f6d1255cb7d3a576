#!/usr/bin/env python3
"""bench.py — Gbp/s of the minimizer scan (k=31, w=11) over 150-bp reads on N MI355X GPUs.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  A step = one pass of the fused minimizer scan
(bl_scan_minimizers, canonical 31-mers, window 11, seed 42) over this rank's whole shard
(BASELINE.json configs[2]: 50 Gbp of 150-bp reads per GPU, synthetic, resident in HBM before the
timed region), issued as consecutive <= 1.5 Gbp ranges whose records (value, position, hash) are
materialised into HBM output arrays.  Shards are independent (weak scaling, no data-path collective);
the only collective is the optional count reduction, done after the timed region over RCCL.

Prints ONE JSON line on rank 0, including
  roofline     achieved HBM-read GB/s of the scan kernel (1 byte/base x bases per launch / mean
               kernel time from HIP events on the launch stream) against the 8 TB/s peak
  cpu_baseline the CPU oracle (port of the reference algorithm) timed on this box's host cores on a
               bounded sample of the same reads, plus a bit-exact check of the GPU result on it
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

UNIT, W, SEED, READ_LEN = 31, 11, 42, 150
N_CU = 256
CLOCK_HZ = 2.4e9
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gbp", type=float, default=50.0, help="Gbp per GPU (BASELINE config: 50)")
    ap.add_argument("--chunk-reads", type=int, default=10_000_000, help="reads per scan range (<= 2^31 bases)")
    ap.add_argument("--lanes", type=int, default=2, choices=(1, 2),
                    help="execution lanes of the context: 2 = the record pass of one range runs beside the hashing pass of the next")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import biolib_amd
    from biolib_amd import FLAG_CANONICAL, Result

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearse = world > 1 and os.environ.get("BL_BENCH_REHEARSE") == "1"
    coll_dev = "cpu" if rehearse else "cuda"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            # rehearsal of the multi-rank flow on a ONE-GPU box (never a measurement): every rank on cuda:0, gloo for
            # the bookkeeping collectives because RCCL refuses two ranks on one device
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm
    n_gpus = world
    if args.gpus != world and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; running {world} rank(s)", file=sys.stderr)

    dev = local_rank if (world > 1 and not rehearse) else 0
    torch.cuda.set_device(dev)
    ctx = biolib_amd.Context(dev, torch_stream=False, lanes=args.lanes)  # own streams; outputs below are double-buffered

    n_reads = int(args.gbp * 1e9) // READ_LEN
    n_bases = n_reads * READ_LEN
    batch = ctx.synth(SEED + rank, n_bases, READ_LEN)  # per-shard seed: the N-GPU input is N distinct shards

    chunk = min(args.chunk_reads, n_reads) * READ_LEN
    assert chunk <= 2**31
    ranges = [(a, min(chunk, n_bases - a)) for a in range(0, n_bases, chunk)]
    cap = int(chunk * 2.25 / (W + 1)) + 65536  # expected density 2/(w+1) = 0.167 rec/window; ~0.128 rec/base on 150-bp reads
    outs = [(ctx.empty_u64(cap), ctx.empty_u64(cap), ctx.empty_u64(cap)) for _ in range(2)]  # double-buffered record arrays
    flags = FLAG_CANONICAL

    def one_step(results):
        for i, (a, n) in enumerate(ranges):
            v, p, h = outs[i & 1]
            r = Result()
            batch.minimizers_raw(UNIT, W, SEED, flags, first=a, n=n, values=v, positions=p, hashes=h, capacity=cap, result=r)
            results.append(r)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = []
        one_step(res)
        ctx.sync()
        for r in res:
            assert r.status == 0, f"scan failed with status {r.status} (count {r.count}, capacity {cap})"

    ctx.kernel_timing(True)
    barrier()
    t0 = time.perf_counter()
    all_res = []
    for _ in range(args.steps):
        one_step(all_res)
    ctx.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = ctx.kernel_time()
    ctx.kernel_timing(False)
    for r in all_res:
        assert r.status == 0, f"scan failed with status {r.status}"
    step_res = all_res[-len(ranges):]
    count = sum(int(r.count) for r in step_res)
    xor_hash = 0
    for r in step_res:
        xor_hash ^= int(r.xor_hash)
    # every timed step scanned the same shard: its ranges must report the same counts and digests each time
    for k in range(args.steps - 1):
        for a, b in zip(all_res[k * len(ranges):(k + 1) * len(ranges)], step_res):
            assert (a.count, a.xor_hash, a.xor_pos, a.xor_value) == (b.count, b.xor_hash, b.xor_pos, b.xor_value), "steps disagree"

    t_max = elapsed
    total_count = count
    allreduce_ms = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_max = float(t.item())
        # optional final count reduction over RCCL/xGMI (64-bit sums all-reduced, XOR digests gathered and
        # folded: biolib_amd/shard.py); outside the timed region, its time is reported
        from biolib_amd.shard import reduce_digests

        torch.cuda.synchronize()
        ta = time.perf_counter()
        tot = reduce_digests(dict(count=count, xor_hash=xor_hash), device=coll_dev)
        torch.cuda.synchronize()
        allreduce_ms = (time.perf_counter() - ta) * 1e3
        total_count, xor_hash = tot["count"], tot["xor_hash"]
        k = torch.tensor([kernel_ms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        kernel_ms = float(k.item())

    out = None
    if rank == 0:
        total_bases = float(n_bases) * n_gpus * args.steps
        value = total_bases / t_max / 1e9
        bases_per_launch = sum(n for _, n in ranges) / len(ranges)
        avg_kernel_s = kernel_ms / 1e3 / max(launches, 1)
        achieved = bases_per_launch * 1.0 / avg_kernel_s / 1e9  # 1 algorithmic byte per base
        traffic = None
        valu = None
        hbm_actual = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")  # written from a separate rocprofv3 --pmc run, see DESIGN.md
        if os.path.exists(tp):
            try:
                prof = json.load(open(tp))
                traffic = int(prof["hbm_bytes_per_base"] * bases_per_launch)  # PMC bytes per base x this run's bases per launch
                # what actually moves through HBM per range (both passes, PMC) over the time a range takes in this run
                emit_bytes = prof.get("emit_kernel", {}).get("hbm_bytes_per_launch_measured", 0) / prof["measured_bases_per_launch"] * bases_per_launch
                range_s = t_max / args.steps / len(ranges)
                hbm_actual = {"bytes_per_range_both_passes": int(traffic + emit_bytes), "GBps": round((traffic + emit_bytes) / range_s / 1e9, 1),
                              "frac_of_8TBps": round((traffic + emit_bytes) / range_s / 1e9 / HBM_PEAK_GBPS, 4)}
                # second ceiling: VALU issue.  A wave64 instruction occupies a SIMD16 for 4 cycles, so the
                # nominal peak is CUs x 4 SIMDs x clock / 4 wave-instructions per second (some simple ops retire faster,
                # which is how the fraction can pass 1).
                per_base = prof["valu_wave_instr_per_base"]
                if args.lanes == 2:  # the record pass of the previous range shares the SIMDs with this kernel: count its instructions too
                    per_base += prof.get("emit_valu_wave_instr_per_base", 0.0)
                winstr = per_base * bases_per_launch
                peak_winstr = N_CU * 4 * CLOCK_HZ / 4.0
                valu = {"wave_instr_per_launch": int(winstr), "achieved_Ginstr_s": round(winstr / avg_kernel_s / 1e9, 1),
                        "peak_Ginstr_s": round(peak_winstr / 1e9, 1), "frac": round(winstr / avg_kernel_s / peak_winstr, 3),
                        "lane_instr_per_base": round(per_base * 64, 1),
                        "kernels": "scan_count + co-running scan_emit" if args.lanes == 2 else "scan_count",
                        "source": "SQ_INSTS_VALU, profiles/r01_pmc_summary.txt; 256 CUs x 4 SIMD16 at 2.4 GHz"}
            except Exception:
                traffic = None
        out = {
            "metric": "Gbp/s minimizer-scanned (k=31,w=11)",
            "value": round(value, 3),
            "unit": "Gbp/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(t_max / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": f"minimizer_view k=31 w=11 (canonical 31-mer units, window 11, seed 42) over {args.gbp:g} Gbp of 150 bp short reads per GPU",
                "bases_per_gpu": n_bases, "read_len": READ_LEN, "reads_per_gpu": n_reads, "ranges_per_step": len(ranges),
                "bases_per_launch": int(bases_per_launch), "outputs": "value,position,hash (u64 each) materialised in HBM", "lanes": args.lanes,
                "sharding": f"{n_gpus} independent shard(s), seed 42+rank",
            },
            "records_per_step": total_count,
            "xor_hash": xor_hash,
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": traffic, "kernel": "bl::scan_count_kernel<MODE_MINIMIZER,W=11,U=31,C=1> (pass 1 of 2)", "avg_kernel_ms": round(avg_kernel_s * 1e3, 4),
                "launches_timed": launches, "algorithmic_bytes_per_launch": int(bases_per_launch),
                "note": "integer-ALU bound before HBM: 6 x 64-bit multiplies per base (MurmurHash3_x64_128), see DESIGN.md"
                        + ("; with 2 lanes this kernel's duration includes sharing the SIMDs with the previous range's scan_emit_kernel "
                           "(alone: --lanes 1, profiles/r01_bench_1gpu_lanes1.json)" if args.lanes == 2 else ""),
            },
        }
        if rehearse:
            out["rehearsal"] = "all ranks on cuda:0 over gloo: flow check only, NOT a measurement"
        if valu is not None:
            out["roofline"]["valu"] = valu
        if hbm_actual is not None:
            out["roofline"]["hbm_actual"] = hbm_actual
        if n_gpus == 1:
            try:
                rd, cp = ctx.probe_hbm(8 << 30, 5)
                out["roofline"]["peak_measured"] = {"read_GBps": round(rd, 1), "copy_GBps": round(cp, 1),
                                                    "note": "this device, 8 GiB buffers: read-only stream kernel / DtoD copy (read+write bytes)"}
            except Exception as e:  # a probe failure must not lose the bench line
                out["roofline"]["peak_measured"] = {"error": str(e)}
        if allreduce_ms is not None:
            out["count_allreduce_ms"] = round(allreduce_ms, 3)
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(np, ctx, batch, n_bases)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def usable_cores():
    """threads worth starting: the affinity mask, capped by the cgroup CPU quota when there is one"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(np, ctx, batch, n_bases):
    """The CPU oracle (oracle/bl_oracle.c, streaming variant = the reference's operation counts:
    per-base roll, one MurmurHash3_x64_128 per 31-mer, ring-buffer window minimum) on the first
    reads of the same synthetic workload.  Also re-checks the GPU result on that sample bit-exactly."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    cores = usable_cores()
    n1 = min(n_bases, 2_000_000 * READ_LEN)       # 300 Mbp single-thread (~8 s)
    nall = min(n_bases, 20_000_000 * READ_LEN)    # 3 Gbp on all cores
    seq = batch.download(0, nall)
    offs = np.arange(0, nall + 1, READ_LEN, dtype=np.uint64)
    t = time.perf_counter()
    d1 = O.minimizer_digest(seq[:n1], offs[: n1 // READ_LEN + 1], UNIT, W, SEED, True, threads=1)
    t1 = time.perf_counter() - t
    t = time.perf_counter()
    dall = O.minimizer_digest(seq, offs, UNIT, W, SEED, True, threads=cores)
    tall = time.perf_counter() - t
    from biolib_amd import FLAG_CANONICAL, FLAG_SYNC

    gc = gv = gh = gp = 0
    step = 10_000_000 * READ_LEN  # scan ranges hold at most 2^31 positions
    for a in range(0, nall, step):
        g = batch.minimizers_raw(UNIT, W, SEED, FLAG_CANONICAL | FLAG_SYNC, first=a, n=min(step, nall - a))
        gc += int(g.count); gv ^= int(g.xor_value); gh ^= int(g.xor_hash); gp ^= int(g.xor_pos)
    same = (gc, gv, gh, gp) == (dall["count"], dall["xor_value"], dall["xor_hash"], dall["xor_pos"])
    return {
        "value": round(nall / tall / 1e9, 4), "unit": "Gbp/s", "cores": cores, "kind": "port",
        "sample": f"first {nall // READ_LEN} reads ({nall / 1e9:.2f} Gbp) of rank 0's shard, OpenMP over reads, {cores} threads; "
                  f"single thread on the first {n1 / 1e9:.2f} Gbp",
        "single_thread_value": round(n1 / t1 / 1e9, 4),
        "gpu_result_bit_identical_on_sample": bool(same),
        "sample_records": dall["count"],
    }


if __name__ == "__main__":
    main()
