#!/usr/bin/env python3
"""bench.py — Gbp/s of the minimizer scan (k=31, w=11) over 150-bp reads on N MI355X GPUs.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  A step = one pass of the fused minimizer scan
(bl_scan_minimizers, canonical 31-mers, window 11, seed 42) over this rank's whole shard
(BASELINE.json configs[2]: 50 Gbp of 150-bp reads per GPU, synthetic, resident in HBM before the
timed region), issued as consecutive <= 0.75 Gbp ranges whose records (value, position, hash) are
materialised into HBM output arrays.  Shards are independent (weak scaling, no data-path collective);
the only collective is the optional count reduction, done after the timed region over RCCL.

Prints ONE JSON line on rank 0, including
  roofline       achieved HBM-read GB/s of the dominant kernel (1 byte/base x bases per launch / mean kernel time from
                 HIP events on the launch stream) against the 8 TB/s peak; `valu` = the second ceiling: the instructions
                 the kernels execute (ISA census x PMC count) priced at the issue cycles tools/ubench_valu.hip measured,
                 over the SIMD cycles available at the shader clock measured DURING the timed region
  other_configs  (1 GPU) one pass each over the other BASELINE.json configurations at their stated sizes:
                 C2 canonical 31-mer + hash64 over 10 Gbp, C4 super-k-mers k=31 m=15 over 50 Gbp of 10-kbp reads,
                 C5 syncmers k=31 s=11 over 50 Gbp of 10-kbp reads — Gbp/s, kernel ms, roofline fraction
  next_rows      (1 GPU) the rows SURVEY.md §8(f) marks "next", each on a small workload: the partitioned k-mer counter's
                 chain (scan -> records -> owner split -> count) in Gbp/s, the device BGZF inflate in GB/s of text and a
                 plain gzip file (one deflate stream per member, decoded in parts by the host's cores) to device batches
  cpu_baseline   the CPU oracle (port of the reference algorithm) timed on this box's host cores on a bounded sample
                 of the same reads, a bit-exact check of the GPU result on it, and the reference itself on C2's path
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

UNIT, W, SEED, READ_LEN = 31, 11, 42, 150
N_SIMD = 256 * 4
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MODEL_PATH = os.path.join(ROOT, "profiles", "valu_model.json")   # tools/valu_model.py
TRAFFIC_PATH = os.path.join(ROOT, "profiles", "traffic.json")    # tools/summarise_profiles.py


KERNEL_SOURCES = ("bl_kernels.hip", "bl_scan_core.hpp", "bl_scan_phases.hpp", "bl_scan_frl.hpp", "bl_launch.hpp")  # what the scan kernels are compiled from


def kernel_sources_digest(root=ROOT):
    """sha256 over the scan kernels' sources.  tools/summarise_profiles.py stamps profiles/traffic.json and valu_model.json with
    it when it turns a collection run into those files; this run recomputes it: a difference means the counters the roofline
    line quotes were measured on OTHER kernels than the ones that ran (`traffic_stale`, `valu.stale`)."""
    import hashlib

    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(name.encode())
        with open(os.path.join(root, "biolib_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def profile_is_stale(provenance, digest):
    """True unless the profile carries the digest of the sources as they are now (a profile without one is of unknown age: stale)."""
    return (provenance or {}).get("kernel_sources_sha256") != digest


def load_json_or_none(path):
    """A missing profile file is reported as such in the JSON line; a broken one is an error, not a silent null."""
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def valu_ceiling(model, kernels, bases, seconds, ghz):
    """Second ceiling: SIMD cycles the named kernels need per `bases` at the measured issue cost of their
    instructions, over the SIMD cycles `seconds` hold at the measured shader clock.  The issue costs are the sustained rates
    tools/ubench_valu.hip measures per opcode in isolation (8 waves per SIMD, nothing else running); a real mix pairs its cheap and
    its expensive instructions a little better than any single-opcode loop, so the fraction is good to a few percent and a kernel
    that issues flat out can read 1.01 (the closed-syncmer scan does)."""
    if model is None:
        return {"error": "profiles/valu_model.json missing: run tools/collect_profiles.sh on the GPU box and tools/valu_model.py"}
    need, parts = 0.0, {}
    # the kernels that decide listed tiles a second time run inside the same region: counted where the model holds them
    kernels = list(kernels) + [k.replace("_count", "_redo") for k in kernels if k.endswith("_count") and k.replace("_count", "_redo") in model["kernels"]]
    for k in kernels:
        km = model["kernels"][k]
        if "cycles_per_base" not in km:
            continue
        need += km["cycles_per_base"] * bases
        parts[k] = {"valu_lane_instr_per_base": round(km["valu_per_base"] * 64, 1), "simd_cycles_per_base": round(km["cycles_per_base"], 4),
                    "avg_issue_cycles": round(km["cycles_per_base"] / km["valu_per_base"], 3)}
    avail = N_SIMD * ghz * 1e9 * seconds
    return {"frac": round(need / avail, 4), "stale": profile_is_stale(model["provenance"].get("pmc"), kernel_sources_digest()), "simd_cycles_needed": int(need), "simd_cycles_available": int(avail), "shader_clock_GHz": round(ghz, 3),
            "kernels": parts, "issue_cycles": model["issue_cycles"], "source": model["provenance"]}


def kernel_time_fits(avg_kernel_ms, launches, steps, ms_per_step, slack=1.01):
    """The timed kernel's launches are serialised on the device (one stream, or two lanes whose pass-1 kernels wait for each other), so
    their summed duration cannot exceed the wall time of the steps they ran in.  A figure that does was measured while two instances
    shared the chip: each event pair then spans a stretched duration and the roofline fraction derived from it means nothing."""
    return avg_kernel_ms * launches / max(steps, 1) <= ms_per_step * slack


class PowerSampler:
    """The cards' hwmon sensors (power, shader clock) sampled every 50 ms beside the timed region — the headline scan sits on the card's
    power limit, and the line should say what the limit was in THIS run.  The host's other cards show in hwmon too: the
    summary is of the card that drew the most on average.  Never fatal: no sensors, no `power` object."""

    def __init__(self, root="/sys/class/drm"):
        import glob
        import threading

        self.cards = {}
        for hw in glob.glob(os.path.join(root, "card*/device/hwmon/hwmon*")):
            files = {n: os.path.join(hw, n) for n in ("power1_input", "power1_average", "power1_cap", "freq1_input") if os.path.exists(os.path.join(hw, n))}
            if "power1_input" in files or "power1_average" in files:
                self.cards[os.path.relpath(hw, root).split(os.sep)[0]] = files
        self.samples = []
        self.stop = False
        self.thread = threading.Thread(target=self._loop, daemon=True) if self.cards else None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return int(f.read().strip())
        except Exception:
            return None

    def _loop(self):
        while not self.stop:
            self.samples.append({c: {n: self._read(p) for n, p in files.items()} for c, files in self.cards.items()})
            time.sleep(0.05)

    def start(self):
        if self.thread:
            self.thread.start()
        return self

    def finish(self):
        if not self.thread:
            return None
        self.stop = True
        self.thread.join()
        best, best_mean = None, -1.0
        for c in self.cards:
            v = [s[c].get("power1_input") or s[c].get("power1_average") for s in self.samples]
            v = [x for x in v if x is not None]
            if v and sum(v) / len(v) > best_mean:
                best, best_mean = c, sum(v) / len(v)
        if best is None or len(self.samples) < 3:
            return None
        pw = sorted(x for x in ((s[best].get("power1_input") or s[best].get("power1_average")) for s in self.samples) if x is not None)
        fq = sorted(x for x in (s[best].get("freq1_input") for s in self.samples) if x is not None)
        cap = next((s[best].get("power1_cap") for s in self.samples if s[best].get("power1_cap")), None)
        q = lambda a, f: a[int(f * (len(a) - 1))] if a else None
        return {"card": best, "samples": len(pw), "power_W": {"median": round(q(pw, 0.5) / 1e6), "p90": round(q(pw, 0.9) / 1e6), "max": round(pw[-1] / 1e6)},
                "power_cap_W": round(cap / 1e6) if cap else None, "sensor_clock_MHz": {"median": round(q(fq, 0.5) / 1e6), "p10": round(q(fq, 0.1) / 1e6)} if fq else None,
                "note": "hwmon of the busiest card, every 50 ms over the timed region (steps shorter than a few samples say little)"}


def free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, child_cmd, env=None, poll_s=0.2):
    """Start `n` fresh rank processes of `child_cmd` on this node (RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set as torch.distributed.run would set them), relay rank 0's stdout (the one JSON line) to ours and everything else
    to stderr, and return the WORST exit code.  When a rank fails, the ranks it started — those exact PIDs — are terminated, so that
    nobody waits in a collective for ever.  The caller has not touched the GPU: the ranks are children, never an exec of this process
    (one stream + one host thread per GPU in SURVEY.md §7 step 5 is one process per GPU here; the reference's own drivers are one
    process, tests/test_kmer_view.cpp:23-42)."""
    import subprocess

    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("MASTER_PORT", str(free_port()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), GROUP_RANK="0")
        procs.append(subprocess.Popen(child_cmd, env=e, stdout=None if r == 0 else sys.stderr))
    codes = [None] * n
    try:
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
            if any(c not in (None, 0) for c in codes):
                break
            time.sleep(poll_s)
    finally:
        ended_here = []
        for i, p in enumerate(procs):  # only after a failure or an interrupt is anything still running here
            if codes[i] is None and p.poll() is None:
                p.terminate()
                ended_here.append(i)
        for i, p in enumerate(procs):
            if codes[i] is None:
                try:
                    codes[i] = p.wait(timeout=30)
                except subprocess.TimeoutExpired:
                    p.kill()
                    codes[i] = p.wait()
    worst = 0
    for i, c in enumerate(codes):
        if c != 0 and i not in ended_here:  # the ranks this function ended are a consequence, not the failure
            worst = max(worst, c if c > 0 else 128 - c)  # a rank ended by signal s reads as 128 + s
    return worst or (1 if any(codes) else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gbp", type=float, default=50.0, help="Gbp per GPU (BASELINE config: 50)")
    ap.add_argument("--chunk-reads", type=int, default=5_000_000,
                    help="reads per scan range (<= 2^31 bases).  5 M reads = 0.75 Gbp: 525-527 Gbp/s against 512-515 with ranges of 1.5 Gbp, three rounds on one box, "
                         "the card drawing 1,370 instead of 1,310 W (profiles/r04_ab_summary.txt, DESIGN.md section 6.3); below 2.5 M reads the launches show")
    ap.add_argument("--lanes", type=int, default=2, choices=(1, 2),
                    help="execution lanes of the context: 2 = the record pass of one range runs beside the hashing pass of the next")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-next-rows", action="store_true")
    ap.add_argument("--no-other-shapes", action="store_true", help="skip the 151-bp companion of the headline in other_configs (the PMC passes of tools/collect_profiles.sh: "
                                                                   "its record kernel is the headline's, and a counter mean should belong to one workload)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the upload-inclusive companion figure")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="context switch for A/B runs (bl_ctx_set_option: exact_windows, position_tiled, emit_lds_bytes); repeatable; named in the JSON line")
    ap.add_argument("--other-gbp", type=float, default=0.0, help="size of the other configs (0 = as BASELINE.json states them: 10 / 50 / 50 Gbp)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # invoked without a launcher (`python bench.py --gpus N`): this process becomes the launcher, BEFORE anything touches the GPU
        if os.environ.get("BL_BENCH_REHEARSE") != "1":
            import torch  # device_count() reads the driver's node list, it does not initialise the GPU

            have = torch.cuda.device_count()
            if have < args.gpus:
                print(f"bench.py: --gpus {args.gpus} but {have} device(s) visible: not running a {have}-rank line under an {args.gpus}-GPU label", file=sys.stderr)
                sys.exit(2)
        sys.exit(launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

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
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s): refusing to print a line whose n_gpus is not what was asked for", file=sys.stderr)
        sys.exit(2)
    if world > 1 and not rehearse and torch.cuda.device_count() < world:
        print(f"bench.py: rank {rank}: {world} ranks but {torch.cuda.device_count()} device(s) visible", file=sys.stderr)
        sys.exit(2)
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

    dev = local_rank if (world > 1 and not rehearse) else 0
    torch.cuda.set_device(dev)
    ctx = biolib_amd.Context(dev, torch_stream=False, lanes=args.lanes)  # own streams; outputs below are double-buffered
    opts = {}
    for item in args.opt:
        name, _, value = item.partition("=")
        ctx.set_option(name, int(value))
        opts[name] = int(value)

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

    step_est = None
    for _ in range(max(args.warmup, 0)):
        res = []
        tw = time.perf_counter()
        one_step(res)
        ctx.sync()
        step_est = time.perf_counter() - tw
        for r in res:
            assert r.status == 0, f"scan failed with status {r.status} (count {r.count}, capacity {cap})"

    ctx.kernel_timing(True)
    barrier()
    # shader clock over the timed region: one sleeping wave on its own stream, started with the region and stopped at its end
    # (before the barrier's device-wide synchronise, which would otherwise wait for it); the duration is only an upper bound
    probe = None if os.environ.get("BL_NO_CLOCK_PROBE") else ctx.clock_probe_start(5000)  # (switch: A/B runs under a profiler)
    sampler = None
    if rank == 0 and world == 1:
        try:
            sampler = PowerSampler().start()
        except Exception:
            sampler = None
    t0 = time.perf_counter()
    all_res = []
    ctx.mark()  # markers on the device's timeline behind every step: per-step times without a sync between the steps
    for _ in range(args.steps):
        one_step(all_res)
        ctx.mark()
    ctx.sync()
    clock_ghz = ctx.clock_probe_finish(probe) if probe is not None else 2.4
    barrier()
    elapsed = time.perf_counter() - t0
    power = None
    if sampler is not None:
        try:
            power = sampler.finish()
        except Exception:
            power = None
    marks = ctx.mark_times()
    step_ms = [b - a for a, b in zip(marks[:-1], marks[1:])]
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
    rank_ms = [elapsed / args.steps * 1e3]
    c5_reduce = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_max = float(t.item())
        # per-step times: the slowest rank's; per-rank whole-run times: a straggler shows as max >> min
        sm = torch.tensor(step_ms, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(sm, op=dist.ReduceOp.MAX)
        step_ms = [float(x) for x in sm.cpu().tolist()]
        every = [torch.zeros(1, dtype=torch.float64, device=coll_dev) for _ in range(world)]
        dist.all_gather(every, torch.tensor([elapsed / args.steps * 1e3], dtype=torch.float64, device=coll_dev))
        rank_ms = [float(x.item()) for x in every]
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
        c5_reduce = c5_count_reduce(ctx, torch, dist, rank, world, coll_dev)

    out = None
    if rank == 0:
        model = load_json_or_none(MODEL_PATH)
        prof = load_json_or_none(TRAFFIC_PATH)
        total_bases = float(n_bases) * n_gpus * args.steps
        value = total_bases / t_max / 1e9
        step_values = sorted(float(n_bases) * n_gpus / (ms / 1e3) / 1e9 for ms in step_ms if ms > 0)
        bases_per_launch = sum(n for _, n in ranges) / len(ranges)
        avg_kernel_s = kernel_ms / 1e3 / max(launches, 1)
        achieved = bases_per_launch * 1.0 / avg_kernel_s / 1e9  # 1 algorithmic byte per base
        range_s = t_max / args.steps / len(ranges)
        traffic, hbm_actual, traffic_note = None, None, "profiles/traffic.json missing: no PMC traffic figure"
        traffic_stale = True
        if prof is not None:
            traffic_stale = profile_is_stale(prof.get("provenance"), kernel_sources_digest())
            c3 = prof["c3"]
            traffic = int(c3["count_kernel"]["hbm_bytes_per_base"] * bases_per_launch)  # PMC bytes per base x this run's bases per launch
            both = (c3["count_kernel"]["hbm_bytes_per_base"] + c3["emit_kernel"]["hbm_bytes_per_base"]) * bases_per_launch
            hbm_actual = {"bytes_per_range_both_passes": int(both), "GBps": round(both / range_s / 1e9, 1), "frac_of_8TBps": round(both / range_s / 1e9 / HBM_PEAK_GBPS, 4)}
            traffic_note = prof["provenance"]
        # the kernel the events bracketed, by the path the options select (launch_count_frl / launch_count_mode in bl_kernels.hip)
        if opts.get("position_tiled"):
            kernel_label = "bl::scan_count_kernel<MODE_MINIMIZER,W=11,U=31,C=1> (pass 1 of 2, position-tiled)"
        elif opts.get("exact_windows"):
            kernel_label = "bl::scan_count_frl_kernel<MODE_MINIMIZER,W=11,NS=15,U=31,L=150,C=1> (pass 1 of 2, read-tiled, windows decided on the hashes)"
        else:
            kernel_label = ("bl::scan_count_frl_kernel<MODE_MINIMIZER,W=11,NS=15,U=31,L=150,C=1,APPROX> + scan_redo_frl_kernel "
                            "(pass 1 of 2, read-tiled, windows decided on murmur64_top; the events bracket both launches)")
        out = {
            "metric": "Gbp/s minimizer-scanned (k=31,w=11)",
            "value": round(value, 3),
            "unit": "Gbp/s",
            "n_gpus": n_gpus,
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "collective_backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if world > 1 else None,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(t_max / args.steps * 1e3, 3),
            "median_value": round(step_values[len(step_values) // 2], 3) if step_values else None,
            "best_value": round(step_values[-1], 3) if step_values else None,
            "step_ms": [round(x, 3) for x in step_ms],
            "step_ms_note": "device timeline markers behind every step (bl_ctx_mark), slowest rank per step; value = all steps / host wall time between the barriers",
            "rank_ms_per_step": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3)},
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
                **({"context_options": opts} if opts else {}),
            },
            "records_per_step": total_count,
            "xor_hash": xor_hash,
            "tiles_decided_again_per_step": {"tiles": int(sum(int(r.redone) for r in step_res)), "of": int(-(-n_bases // 4800)),
                                             "note": "rank 0: tiles of 32 reads in which pass 1, which looks at an approximation of the hash's high dword, met keys too close to order; counted a second time on the hashes (scan_redo_frl_kernel, inside the timed region)"},
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": traffic, "traffic_stale": traffic_stale, "traffic_source": traffic_note,
                "kernel": kernel_label, "avg_kernel_ms": round(avg_kernel_s * 1e3, 4),
                "launches_timed": launches, "algorithmic_bytes_per_launch": int(bases_per_launch),
                "kernel_ms_per_step": round(avg_kernel_s * 1e3 * launches / args.steps, 3),
                "kernel_time_fits_step": kernel_time_fits(avg_kernel_s * 1e3, launches, args.steps, t_max / args.steps * 1e3),
                "note": "VALU-issue bound before HBM: 6 x 64-bit multiplies per 31-mer (MurmurHash3_x64_128), see roofline.valu and DESIGN.md"
                        + ("; with 2 lanes this kernel's duration includes sharing the SIMDs with the previous range's scan_emit_kernel (alone: --lanes 1)"
                           if args.lanes == 2 else ""),
            },
        }
        if rehearse:
            out["rehearsal"] = "all ranks on cuda:0 over gloo: flow check only, NOT a measurement"
        # VALU ceiling over the whole timed region: both passes' instructions against the SIMD cycles a range's share of it holds
        out["roofline"]["valu"] = valu_ceiling(model, ["c3_count", "c3_emit"], bases_per_launch, range_s, clock_ghz)
        if hbm_actual is not None:
            out["roofline"]["hbm_actual"] = hbm_actual
        if power is not None:
            out["power"] = power
        if n_gpus == 1:
            try:
                rd, cp = ctx.probe_hbm(8 << 30, 5)
                out["roofline"]["peak_measured"] = {"read_GBps": round(rd, 1), "copy_GBps": round(cp, 1),
                                                    "note": "this device, 8 GiB buffers: read-only stream kernel / DtoD copy (read+write bytes)"}
                out["roofline"]["frac_of_measured_read_peak"] = round(achieved / rd, 5)
            except Exception as e:  # a probe failure must not lose the bench line
                out["roofline"]["peak_measured"] = {"error": str(e)}
        if allreduce_ms is not None:
            out["count_allreduce_ms"] = round(allreduce_ms, 3)
        if c5_reduce is not None:
            out["c5_count_reduce"] = c5_reduce
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(np, ctx, batch, n_bases)
        if n_gpus == 1 and not args.no_h2d:
            try:
                out["h2d_inclusive_Gbps"] = h2d_inclusive(np, torch, biolib_amd, dev, batch, cap)
            except Exception as e:  # the companion figure must not lose the bench line
                out["h2d_inclusive_Gbps"] = {"error": repr(e)[:300]}
    if n_gpus == 1 and not args.no_other_configs:
        batch.close()
        del outs
        torch.cuda.empty_cache()
        out["other_configs"] = other_configs(ctx, args, load_json_or_none(MODEL_PATH))
        if not args.no_next_rows:
            out["next_rows"] = next_rows(ctx)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def h2d_inclusive(np, torch, biolib_amd, dev, batch, cap):
    """The upload-INCLUSIVE rate of the same scan ("end-to-end incl. H2D reported separately"; the reference's
    drivers start from host memory, tests/test_kmer_view.cpp:30-42): 8.25 Gbp of the shard's reads leave page-locked host
    memory in eleven chunks through bl_batch_upload_reads and are scanned as they arrive, records materialised as in the
    headline.  Double-buffered over two contexts: the upload of one chunk (synchronous, as the ABI promises the caller its
    buffer back) runs beside the scan of the chunk before it.  Never `value`: inputs of the headline are resident."""
    n_chunk = 5_000_000 * READ_LEN
    n_chunks = 11
    n_chunk = min(n_chunk, batch.n_bases // READ_LEN * READ_LEN)
    host = torch.empty(n_chunk, dtype=torch.uint8).pin_memory()
    host.numpy()[:] = batch.download(0, n_chunk)
    ctxs = [biolib_amd.Context(dev, torch_stream=False, lanes=1) for _ in range(2)]
    outs = [tuple(c.empty_u64(cap) for _ in range(3)) for c in ctxs]
    live = [None, None]
    res = []

    def run(k):
        for i in range(k):
            j = i & 1
            if live[j] is not None:
                live[j].close()  # waits for its scan, long finished: it ran beside the other context's upload
            live[j] = ctxs[j].upload(host.numpy(), read_len=READ_LEN)
            r = biolib_amd.Result()
            v, p, h = outs[j]
            live[j].minimizers_raw(UNIT, W, SEED, biolib_amd.FLAG_CANONICAL, values=v, positions=p, hashes=h, capacity=cap, result=r)
            res.append(r)
        for c in ctxs:
            c.sync()

    run(2)  # buffers allocated, pages touched
    res.clear()
    t0 = time.perf_counter()
    run(n_chunks)
    dt = time.perf_counter() - t0
    ok = all(r.status == 0 and r.count == res[0].count and r.xor_hash == res[0].xor_hash for r in res)
    for b in live:
        if b is not None:
            b.close()
    for c in ctxs:
        c.close()
    return {"value": round(n_chunk * n_chunks / dt / 1e9, 2), "unit": "Gbp/s", "bases": n_chunk * n_chunks, "chunks": n_chunks, "host_to_device_GBps": round(n_chunk * n_chunks / dt / 1e9, 2),
            "every_chunk_same_digest": bool(ok),
            "path": "page-locked host memory -> bl_batch_upload_reads -> bl_scan_minimizers (records materialised), two contexts alternating: upload beside scan"}


def c5_count_reduce(ctx, torch, dist, rank, world, coll_dev):
    """BASELINE C5's collective, rehearsed in every multi-rank run: each rank counts the syncmers (k=31, s=11, offsets 0 / 20)
    of its own 1.5-Gbp shard of 10-kbp reads (seed 42 + rank), the counts are summed across the ranks by the path's one
    collective (biolib_amd.shard.reduce_digests: one all-reduce — RCCL over xGMI under the "nccl" backend) and the sum is checked
    against the all-gathered per-rank counts."""
    import biolib_amd as B
    from biolib_amd.shard import reduce_digests

    L = 10_000
    n = 1_500_000_000 // L * L
    b = ctx.synth(SEED + rank, n, L)
    r = b.syncmers_raw(31, 11, 0, 20, 0, B.FLAG_CANONICAL | B.FLAG_SYNC)
    mine = int(r.count)
    b.close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total = reduce_digests(dict(count=mine), device=coll_dev)["count"]
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    every = [None] * world
    dist.all_gather_object(every, mine)
    assert total == sum(every), f"count reduction: {total} != sum of {every}"
    return {"syncmers_all_ranks": int(total), "per_rank": [int(x) for x in every], "allreduce_ms": round(ms, 3), "sum_checked": True, "backend": dist.get_backend()}


def other_configs(ctx, args, model):
    """One pass each (after an untimed one) over the other BASELINE.json configurations on this GPU, at the sizes they are
    stated at: parity-test cases with a driver-timed rate.  Same protocol as the headline: inputs resident, ranges of
    <= 1.5 Gbp, records materialised where the configuration has records, HIP-event time of the dominant kernel."""
    import biolib_amd as B

    CH = 1_500_000_000
    res = {}

    def timed(issue, n_bases, kernels, kernel_name, steps=2, kernel_lanes=None):
        """`value`: wall time of `steps` passes on the context's lanes.  The dominant kernel's duration comes from the same passes when its
        launches are serialised on the device (the window scans: pass 1 of consecutive ranges wait for each other); kernel_lanes=1: a kernel
        that IS the whole scan (C2's dense k-mer kernel) runs beside its own next launch on two lanes — every event pair would then time a
        stretched kernel — so its duration is taken from one more pass on ONE lane, and the roofline fraction with it."""
        issue()
        ctx.sync()
        ctx.kernel_timing(kernel_lanes is None)
        probe = ctx.clock_probe_start(5000)  # stopped by clock_probe_finish below
        t0 = time.perf_counter()
        n_ranges = 0
        for _ in range(steps):
            n_ranges += issue()
        ctx.sync()
        dt = time.perf_counter() - t0
        ghz = ctx.clock_probe_finish(probe)
        k_steps, k_step_ms = steps, dt / steps * 1e3
        if kernel_lanes is None:
            kms, launches = ctx.kernel_time()
        else:
            ctx.set_option("lanes", kernel_lanes)
            ctx.kernel_timing(True)
            t1 = time.perf_counter()
            issue()
            ctx.sync()
            k_steps, k_step_ms = 1, (time.perf_counter() - t1) * 1e3
            kms, launches = ctx.kernel_time()
            ctx.set_option("lanes", args.lanes)
        ctx.kernel_timing(False)
        per_launch = n_bases * k_steps / max(launches, 1)
        k_s = kms / 1e3 / max(launches, 1)
        achieved = per_launch / k_s / 1e9
        fits = kernel_time_fits(k_s * 1e3, launches, k_steps, k_step_ms)
        if not fits:
            print(f"# {kernel_name}: {launches} launches x {k_s * 1e3:.3f} ms do not fit {k_steps} step(s) of {k_step_ms:.3f} ms: kernel figure invalid", file=sys.stderr)
        return {"value": round(n_bases * steps / dt / 1e9, 2), "unit": "Gbp/s", "bases": n_bases, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3), "lanes": args.lanes,
                "kernel": kernel_name, "avg_kernel_ms": round(k_s * 1e3, 4), "launches_timed": launches, "kernel_timed_on_lanes": kernel_lanes if kernel_lanes is not None else args.lanes,
                "kernel_steps": k_steps, "kernel_ms_per_step": round(k_s * 1e3 * launches / k_steps, 3), "kernel_step_ms": round(k_step_ms, 3), "kernel_time_fits_step": fits,
                "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                             **({"valu": valu_ceiling(model, kernels, per_launch, dt / max(n_ranges, 1), ghz)} if kernels else {"shader_clock_GHz": round(ghz, 3)})}}

    # C2: canonical 31-mer 2-bit encode + hash64 digest over 10 Gbp, one sequence (digest only: 80 GB of hashes are not materialised)
    n = int((args.other_gbp or 10.0) * 1e9)
    b = ctx.synth(SEED, n)

    def c2():
        k = 0
        for a in range(0, n, CH):
            b.kmers_raw(31, 0, B.FLAG_CANONICAL, first=a, n=min(CH, n - a))
            k += 1
        return k

    res["C2_kmer_hash_10Gbp"] = timed(c2, n, ["c2_kmer"], "bl::kmer_kernel", kernel_lanes=1)
    b.close()

    # beside the headline shape: the same scan on 151-bp reads (what current sequencers write) — the read-tiled murmur64_top kernel with the read
    # geometry at run time (16 units per lane); same call pattern as the headline (ranges of 5 M reads, records materialised)
    if not args.no_other_shapes:
        L = 151
        n = int((args.other_gbp or 50.0) * 1e9) // L * L
        b = ctx.synth(SEED, n, L)
        chunk = 5_000_000 * L
        cap3 = int(chunk * 2.25 / (W + 1)) + 65536
        bufs = [(ctx.empty_u64(cap3), ctx.empty_u64(cap3), ctx.empty_u64(cap3)) for _ in range(2)]

        def c3b():
            k = 0
            for i, a in enumerate(range(0, n, chunk)):
                v, p, h = bufs[i & 1]
                b.minimizers_raw(UNIT, W, SEED, B.FLAG_CANONICAL, first=a, n=min(chunk, n - a), values=v, positions=p, hashes=h, capacity=cap3)
                k += 1
            return k

        res["C3_shape_on_151bp_reads_50Gbp"] = timed(c3b, n, [], "bl::scan_count_frl_kernel<MODE_MINIMIZER,W=11,NS=16,U=31,L=run time,C=1,APPROX> + scan_redo_frl_kernel")
        b.close()
        del bufs

    # C4 / C5: 50 Gbp of 10-kbp reads
    L = 10_000
    n = int((args.other_gbp or 50.0) * 1e9) // L * L
    b = ctx.synth(SEED, n, L)
    chunk = CH // L * L
    cap4 = int(chunk * 2.3 / 18) + 65536
    bufs = [(ctx.empty_u64(cap4), ctx.empty_u64(cap4), ctx.empty_u8(cap4), ctx.empty_u8(cap4), ctx.empty_u64(cap4)) for _ in range(2)]

    def c4():
        k = 0
        for i, a in enumerate(range(0, n, chunk)):
            mn, fp, mp, sz, hs = bufs[i & 1]
            b.super_kmers_raw(31, 15, SEED, B.FLAG_CANONICAL, first=a, n=min(chunk, n - a), minimizers=mn, first_pos=fp, mm_pos=mp, sizes=sz, hashes=hs, capacity=cap4)
            k += 1
        return k

    res["C4_super_kmers_50Gbp_10kbp_reads"] = timed(c4, n, ["c4_count", "c4_emit"], "bl::scan_count_kernel<MODE_SUPERKMER,W=17,U=15,C=1>")
    del bufs
    cap5 = int(chunk * 2.6 / 21) + 65536
    pbuf = [ctx.empty_u64(cap5) for _ in range(2)]

    def c5():
        k = 0
        for i, a in enumerate(range(0, n, chunk)):
            b.syncmers_raw(31, 11, 0, 20, 0, B.FLAG_CANONICAL, first=a, n=min(chunk, n - a), positions=pbuf[i & 1], capacity=cap5)
            k += 1
        return k

    res["C5_syncmers_50Gbp_shard_10kbp_reads"] = timed(c5, n, ["c5_count", "c5_emit"], "bl::scan_count_kernel<MODE_SYNCMER,W=21,U=11,C=1,SY=1> (closed syncmers: sliding minima, no argmin)")
    b.close()
    res["note"] = ("parity of these configurations at these sizes: tests/test_gpu_edges.py (two cuttings agree, 256 Mbp against the oracle); "
                   "C5's 8-GPU RCCL leg needs hardware a 1-GPU box does not have (world-1 nccl reduce is tested)")
    return res


def next_rows(ctx):
    """Driver-timed numbers for the rows SURVEY.md §8(f) marks "next" (they are not the headline and cannot fail it): the
    partitioned k-mer counter's chain on one GPU, the device-side BGZF inflate and the ingest of a plain gzip file, each on a small
    synthetic workload."""
    import struct, zlib
    import numpy as np
    import torch
    import biolib_amd as B

    res = {}
    try:  # scan -> pack -> owner split -> count (tests/perf/count_bench.py is the long form)
        k, m, L = 31, 15, 150
        n = 1_500_000_000 // L * L
        b = ctx.synth(SEED, n, L)
        keys = ctx.empty_u64(int(n * 0.82))
        cnts = torch.empty(int(n * 0.82), dtype=torch.int32, device="cuda")
        best = None
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            recs, hashes = b.super_kmer_records(k, m, seed=SEED, canonical=True)
            bucketed, _ = ctx.partition_records(hashes, recs, 8)
            u, c = ctx.count_super_kmers(bucketed, k, m, seed=SEED, canonical=True, out=(keys, cnts))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res["kmer_count_chain"] = {"value": round(n / best / 1e9, 1), "unit": "Gbp/s", "bases": n, "distinct_kmers": int(u.numel()),
                                   "workload": "1.5 Gbp of 150-bp reads, k=31 m=15 canonical: super-k-mer scan -> 16-byte records -> split by owner (8) -> count in LDS tables"}
        b.close()
        del keys, cnts, recs, hashes, bucketed, u, c
        torch.cuda.empty_cache()
    except Exception as e:  # reported, not fatal
        res["kmer_count_chain"] = {"error": repr(e)[:200]}
    try:  # the same chain on reads of 30x COVERAGE: 0.6 Gbp drawn at random offsets of a 20-Mbp genome (tests/perf/count_coverage_bench.py is the long form)
        k, m, L = 31, 15, 150
        n_reads = 4_000_000
        rng = np.random.default_rng(6)
        genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n_reads * L // 30)]
        pos = rng.integers(0, genome.size - L, n_reads)
        seq = np.ascontiguousarray(genome[pos[:, None] + np.arange(L)[None, :]]).reshape(-1)
        b = ctx.upload(seq, read_len=L)
        best = None
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            recs, hashes = b.super_kmer_records(k, m, seed=SEED, canonical=True)
            u, c = ctx.count_super_kmers(recs, k, m, seed=SEED, canonical=True)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res["kmer_count_chain_30x_coverage"] = {"value": round(seq.size / best / 1e9, 1), "unit": "Gbp/s", "bases": int(seq.size), "distinct_kmers": int(u.numel()),
                                                "kmer_occurrences": int(c.sum()),
                                                "workload": "0.6 Gbp of 150-bp reads drawn at random offsets of a 20-Mbp genome (30x coverage), k=31 m=15 canonical: scan -> records -> count in LDS tables filled in rounds"}
        b.close()
        del recs, hashes, u, c, seq, genome
        torch.cuda.empty_cache()
    except Exception as e:
        res["kmer_count_chain_30x_coverage"] = {"error": repr(e)[:200]}
    try:  # device BGZF inflate: 64 MB of FASTQ text, zlib level 6, 65280-byte members (tests/perf/inflate_bench.py is the long form)
        import ctypes as C
        rng = np.random.default_rng(1)
        n_reads, L = 186_000, 150  # ~1,020 members: one resident set of the kernel (4 per CU)
        seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), (n_reads, L))
        qual = np.where(rng.random((n_reads, L)) < 0.93, ord("F"), rng.choice(np.frombuffer(b":,#", np.uint8), (n_reads, L))).astype(np.uint8)
        text = b"".join(b"@A00123:45:HXXXXXXXX:1:1101:%d:%d 1:N:0:ACGTACGT\n" % (1000 + i % 30000, 1000 + i // 7) + seq[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n"
                        for i in range(n_reads))
        data = bytearray()
        for a in range(0, len(text), 65280):
            chunk = text[a:a + 65280]
            z = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = z.compress(chunk) + z.flush()
            data += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
            data += body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
        data = bytes(data)
        lib = ctx._lib
        cap = len(data) // 26 + 1
        members = np.zeros(cap * 4, np.uint64)
        nm, used, tb = C.c_uint64(), C.c_uint64(), C.c_uint64()
        assert lib.bl_bgzf_walk(data, len(data), 0, 0, members.ctypes.data, cap, C.byref(nm), C.byref(used), C.byref(tb)) == 0
        ptrs = []
        for size in (len(data) + 8, 32 * nm.value, tb.value + 16, 4 * nm.value):
            p = C.c_void_p()
            assert lib.bl_device_alloc(ctx._h, size, C.byref(p)) == 0
            ptrs.append(p)
        lib.bl_copy_to_device(ctx._h, ptrs[0], data, len(data)); lib.bl_copy_to_device(ctx._h, ptrs[1], members.ctypes.data, 32 * nm.value)
        best = None
        for _ in range(4):
            ctx.sync(); t0 = time.perf_counter()
            assert lib.bl_bgzf_inflate(ctx._h, ptrs[0], len(data), ptrs[1], nm.value, ptrs[2], tb.value, ptrs[3]) == 0
            ctx.sync(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out_text = np.zeros(tb.value, np.uint8); st = np.zeros(nm.value, np.uint32)
        lib.bl_copy_to_host(ctx._h, out_text.ctypes.data, ptrs[2], tb.value); lib.bl_copy_to_host(ctx._h, st.ctypes.data, ptrs[3], 4 * nm.value)
        for p in ptrs:
            lib.bl_device_free(ctx._h, p)
        res["bgzf_inflate"] = {"value": round(len(text) / best / 1e9, 2), "unit": "GB/s of text", "members": int(nm.value), "text_MB": round(len(text) / 1e6), "kernel_ms": round(best * 1e3, 3),
                               "same_text_as_zlib": bool(not st.any() and out_text.tobytes() == text),
                               "workload": "FASTQ text (binned qualities), zlib level 6, 65280-byte BGZF members, inflate + CRC-32 on the device"}
    except Exception as e:
        res["bgzf_inflate"] = {"error": repr(e)[:200]}
    try:  # file -> scanned batches, the way the reference's drivers start (tests/test_kmer_view.cpp:23-42 read a file, then iterate): the reader's
          # device path (text spans -> H2D -> parsed on the GPU; BGZF members inflated on the GPU) with the minimizer scan behind every batch
        import tempfile
        times = 24  # the text 24 times over (records and BGZF members concatenate): 1.6 GB of text, 670 Mbp — a reader's start-up (threads, pinned buffers) no longer dominates
        with tempfile.TemporaryDirectory() as d:
            for name, blob in (("plain_fastq_to_scan", text), ("bgzf_file_to_scan", data)):
                path = os.path.join(d, name)
                with open(path, "wb") as f:
                    for _ in range(times):
                        f.write(blob)
                    if name.startswith("bgzf"):
                        f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00")  # BGZF end-of-file marker
                best, ok = None, True
                for _ in range(2):
                    ctx.sync(); t0 = time.perf_counter()
                    r = B.Reader(path)
                    n_seqs = n_b = cnt = 0
                    for batch in r.device_batches(ctx):
                        n_seqs += batch.n_seqs; n_b += batch.n_bases
                        cnt += int(batch.minimizers_raw(UNIT, W, SEED, B.FLAG_CANONICAL | B.FLAG_SYNC).count)
                        batch.close()
                    ctx.sync(); dt = time.perf_counter() - t0
                    r.close()
                    ok = ok and n_seqs == n_reads * times and n_b == n_reads * L * times
                    best = dt if best is None else min(best, dt)
                res[name] = {"value": round(n_reads * L * times / best / 1e9, 2), "unit": "Gbp/s", "file_MB": round(os.path.getsize(path) / 1e6), "text_GB_per_s": round(len(text) * times / best / 1e9, 2),
                             "minimizers": cnt, "reads_and_bases_as_written": bool(ok),
                             "path": "file -> bl_reader_next_batch_device (host reads the file, " + ("members inflated and" if name.startswith("bgzf") else "text") + " parsed on the GPU) -> bl_scan_minimizers per batch, best of 2"}
    except Exception as e:
        res["file_to_scan"] = {"error": repr(e)[:200]}
    try:  # a plain gzip file (ONE deflate stream per member, no BGZF) -> device batches: the host decodes the stream in parts on all its cores
        import tempfile
        packed = zlib.compressobj(6, zlib.DEFLATED, 31)
        packed = packed.compress(text) + packed.flush()
        times = 6  # the member six times over (gzip members concatenate): ~370 MB of text
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "reads.fq.gz")
            with open(path, "wb") as f:
                f.write(packed * times)
            r = B.Reader(path)
            ctx.sync(); t0 = time.perf_counter()
            n_seqs = n_bases = 0
            for batch in r.device_batches(ctx):
                n_seqs += batch.n_seqs; n_bases += batch.n_bases
                batch.close()
            ctx.sync(); dt = time.perf_counter() - t0
            r.close()
        res["gzip_ingest"] = {"value": round(n_bases / dt / 1e9, 2), "unit": "Gbp/s", "text_GB_per_s": round(len(text) * times / dt / 1e9, 2), "host_threads": min(16, os.cpu_count() or 4),
                              "reads_and_bases_as_written": bool(n_seqs == n_reads * times and n_bases == n_reads * L * times),
                              "workload": f"{n_reads * times} reads x {L} bp FASTQ as one `zlib level 6` gzip file of {times} members ({len(packed) * times / 1e6:.0f} MB), decoded by the host's cores in parts, parsed on the device"}
    except Exception as e:
        res["gzip_ingest"] = {"error": repr(e)[:200]}
    return res


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
    n1 = min(n_bases, 2_000_000 * READ_LEN)       # 300 Mbp single-thread (~4 s)
    nall = min(n_bases, 20_000_000 * READ_LEN)    # 3 Gbp on all cores
    seq = batch.download(0, nall)
    offs = np.arange(0, nall + 1, READ_LEN, dtype=np.uint64)
    t = time.perf_counter()
    O.minimizer_digest(seq[:n1], offs[: n1 // READ_LEN + 1], UNIT, W, SEED, True, threads=1)
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
    out = {
        "value": round(nall / tall / 1e9, 4), "unit": "Gbp/s", "cores": cores, "kind": "port",
        "why_port": "the reference's own minimizer_view yields nothing and its minimizer_sampler does not compile (SURVEY.md §3.4, §8a-a7): the reference "
                    "cannot run this configuration; it is timed on the part of the path it can run under reference_c2_anchor",
        "sample": f"first {nall // READ_LEN} reads ({nall / 1e9:.2f} Gbp) of rank 0's shard, OpenMP over reads, {cores} threads; "
                  f"single thread on the first {n1 / 1e9:.2f} Gbp",
        "single_thread_value": round(n1 / t1 / 1e9, 4),
        "gpu_result_bit_identical_on_sample": bool(same),
        "sample_records": dall["count"],
    }
    # anchor: the REFERENCE itself (oracle/_ref, compiled from its unmodified sources) on the part of this path it can run — kmer_view
    # canonical k=31 + hash64 per k-mer (BASELINE C2) — beside the port on the same bases, one thread each (within ~2x)
    if O.have_ref():
        nr = min(n_bases, 100_000_000)
        s = O.as_bytes(seq[:nr])
        t = time.perf_counter()
        rx = int(O.ref().ref_scan_kmer_hash_xor(O._ptr(s), nr, 31, 1, 0))
        tr = time.perf_counter() - t
        t = time.perf_counter()
        px = O.kmer_digest(s, np.array([0, nr], np.uint64), 31, True, 0, drop_last=True, threads=1)
        tp = time.perf_counter() - t
        out["port_vs_reference_single_thread"] = round((nr / tp) / (nr / tr), 3)
        out["port_vs_reference_note"] = ("cpu_baseline.kind is 'port': on the part of the path the reference can run (C2: kmer_view + hash64), one thread, the port's rate "
                                         "divided by the reference's own (below 1: a GPU/CPU ratio read off `value` would flatter the GPU by the inverse; round 3's port stood at 0.64, "
                                         "its hash went through a byte-wise general MurmurHash3 that the reference's compiler had inlined away)")
        out["reference_c2_anchor"] = {"kind": "reference", "path": "kmer_view<uint64_t> canonical k=31 + hash64 per k-mer, `it != cend()` idiom", "cores": 1,
                                      "sample_Mbp": nr // 1_000_000, "reference_Gbps": round(nr / tr / 1e9, 4), "port_Gbps": round(nr / tp / 1e9, 4),
                                      "xor_of_hashes_equal": bool(rx == px["xor_hash"])}
    return out


if __name__ == "__main__":
    main()
