#!/usr/bin/env python3
"""Headline benchmark: AECM 64-sample frames/s (WebRtcAecm_ProcessBlock-equivalents) on MI355X.

One "step" = one pass of the hot path over one batch: a single launch that advances every one of the
S resident streams by T blocks (S*T frames).  Inputs are synthetic 16 kHz far/near pairs generated on
the device and resident in HBM before the timed region; state stays on the device between steps.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--blocks T] [--fs 16000|8000]

N > 1 is launched by torch.distributed.run (one rank per GPU): every rank owns S streams (weak
scaling, no data-path collective); RCCL only gathers the counters.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ALGO_BYTES_PER_FRAME = 384            # 128 far in + 128 near in + 128 out
HBM_PEAK_GBPS = 8000.0                # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def synth_on_device(torch, S, L, seed, device, chunk=8192):
    """Synthetic far/near int16 [S, L] in HBM: white noise x piecewise-constant envelope (0.4 s
    segments from {15..20000}), near = sparse 4-tap echo of far + near-end talk bursts (the recipe
    of webrtc_aecm_amd/synth.py, float math on the GPU; bench data need not be reproducible bit
    for bit across machines, parity is checked elsewhere)."""
    far = torch.empty((S, L), dtype=torch.int16, device=device)
    near = torch.empty((S, L), dtype=torch.int16, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    levels = torch.tensor([15., 60., 500., 3000., 9000., 20000.], device=device)
    talk_levels = torch.tensor([0., 0., 0., 2000., 8000.], device=device)
    seg = 6400
    nseg = L // seg + 2
    taps = ((100, 0.5), (180, -0.3), (333, 0.2), (600, 0.1))
    for s0 in range(0, S, chunk):
        n = min(chunk, S - s0)
        env = levels[torch.randint(0, 6, (n, nseg), generator=g, device=device)].repeat_interleave(seg, dim=1)[:, :L]
        x = torch.randn((n, L), generator=g, device=device) * env * 0.58
        x[:, 1:-1] = (x[:, :-2] + 2 * x[:, 1:-1] + x[:, 2:]) * 0.25
        x = x.clamp_(-32768, 32767).round_()
        echo = torch.zeros_like(x)
        for d, gain in taps:
            echo[:, d:] += gain * x[:, :-d]
        tenv = talk_levels[torch.randint(0, 5, (n, nseg), generator=g, device=device)].repeat_interleave(seg, dim=1)[:, :L]
        y = echo + torch.randn((n, L), generator=g, device=device) * tenv * 0.3
        far[s0:s0 + n] = x.to(torch.int16)
        near[s0:s0 + n] = y.clamp_(-32768, 32767).round_().to(torch.int16)
        del env, x, echo, tenv, y
    return far, near


def usable_cores() -> int:
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(p).read_text().split()
            if p.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(fs, budget_s=12.0):
    """The reference C path (oracle/_ref, kind 'reference') or our restatement of it (kind 'port')
    timed on this box's host cores: one stream per thread, bounded to ~budget_s of wall time."""
    import threading

    from oracle import pyoracle
    from webrtc_aecm_amd.synth import synth_pair
    cores = usable_cores()
    use_ref = pyoracle.have_reference()
    mk = (lambda: pyoracle.RefCoreStream(fs, 1, 1)) if use_ref else (lambda: pyoracle.OracleStream(fs, 1, 1))
    chunk = 4000                                   # blocks per call (~60 ms of CPU work)
    pairs = [synth_pair(100 + i, chunk, fs) for i in range(min(cores, 8))]
    streams = [mk() for _ in range(cores)]
    deadline = [0.0]
    start_gate = threading.Barrier(cores + 1)

    def work(i):
        f, d = pairs[i % len(pairs)]
        s = streams[i]
        start_gate.wait()
        done = 0
        while time.perf_counter() < deadline[0]:
            s.process(f, d)
            done += chunk
        return done
    with ThreadPoolExecutor(max_workers=cores) as ex:
        futs = [ex.submit(work, i) for i in range(cores)]
        deadline[0] = time.perf_counter() + budget_s + 0.05
        start_gate.wait()
        t0 = time.perf_counter()
        total = sum(f.result() for f in futs)
        dt = time.perf_counter() - t0
    return {
        "value": total / dt, "unit": "frames/s", "cores": cores, "kind": "reference" if use_ref else "port",
        "per_core": total / dt / cores,
        "sample": f"{cores} threads (one stream each) x {dt:.1f} s wall = {total} frames of {fs} Hz synthetic pairs, cng on, "
                  f"echoMode 1; " + ("unmodified reference built -O2 from /root/reference (oracle/_ref)" if use_ref
                                     else "oracle/aecm_oracle.c restatement built -O2"),
    }


def load_traffic(workload_key):
    """Per-launch HBM bytes measured with rocprofv3 PMC passes (profiles/hbm_traffic.json, written by
    tools/summarize_profile.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs of this very command,
    FETCH_SIZE calibrated on a known byte count), if recorded for this workload."""
    p = ROOT / "profiles" / "hbm_traffic.json"
    if not p.exists():
        return None
    try:
        rec = json.loads(p.read_text())
        return rec.get(workload_key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def load_issue_note():
    """The resource that actually bounds this kernel (recorded by the last profiling pass)."""
    p = ROOT / "profiles" / "r01_rocprof_summary.json"
    try:
        d = json.loads(p.read_text())["derived"]
        return {"valu_insts_per_frame": round(d["valu_insts_per_frame"], 1),
                "salu_insts_per_frame": round(d["salu_insts_per_frame"], 1),
                "valu_port_busy_frac": round(d["valu_port_busy_frac"], 3),
                "scalar_port_busy_frac": round(d["scalar_port_busy_frac"], 3),
                "model": "one wave64 VALU and one scalar instruction per SIMD per 4 shader cycles",
                "source": "profiles/r01_rocprof_summary.json (rocprofv3 SQ_*), profiles/r01_issue_port_experiments.md"}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=65536, help="streams per GPU (BASELINE config 3: 65536)")
    ap.add_argument("--blocks", type=int, default=128, help="blocks per stream per step (one launch)")
    ap.add_argument("--total-streams", type=int, default=0,
                    help="strong scaling: this many streams in total, split evenly over the GPUs (overrides --streams)")
    ap.add_argument("--clean", action="store_true",
                    help="also feed a clean near-end input (WebRtcAecm_ProcessBlock's nearendClean: third transform per block)")
    ap.add_argument("--fs", type=int, default=16000)
    ap.add_argument("--variant", choices=["fast", "safe"], default="fast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for the counter gather (nccl = RCCL)")
    ap.add_argument("--device", type=int, default=None,
                    help="force this HIP device index on every rank (only for exercising the N>1 code path on a 1-GPU box, "
                         "with --dist-backend gloo)")
    ap.add_argument("--fixed-delay", type=int, default=-1,
                    help="WebRtcAecm_Control fixed delay (>= 0 disables the estimator's choice; 0 = no far-history reads; "
                         "used to calibrate the FETCH_SIZE counter on a known byte count)")
    args = ap.parse_args()

    import torch

    import webrtc_aecm_amd as aecm
    from webrtc_aecm_amd import dist as adist

    rank, local_rank, world = adist.init(args.dist_backend)
    if args.device is not None:
        local_rank = args.device
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the AECM hot path has no CPU implementation in the product")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    S, T, K, W = args.streams, args.blocks, args.steps, args.warmup
    if args.total_streams:
        _, S = adist.shard_range(args.total_streams, rank, world)
    segs = 2
    far, near = synth_on_device(torch, S, segs * T * 64, 1234 + rank, device)
    clean = (near.to(torch.int32) * 3 // 4).to(torch.int16) if args.clean else None
    batch = aecm.AecmBatch(S, args.fs, cng_mode=1, echo_mode=1, device=local_rank,
                           variant=aecm.KERNEL_FAST if args.variant == "fast" else aecm.KERNEL_SAFE)
    if args.fixed_delay >= 0:
        batch.control(args.fixed_delay, 1)
    stride = far.shape[1]

    out_full = torch.empty_like(near)       # same [S][segs*T*64] layout as the inputs

    def step(i):                            # one C-ABI call = one launch = S*T frames
        off = (i % segs) * T * 64 * 2       # byte offset of this step's input segment
        batch.process_device(far.data_ptr() + off, near.data_ptr() + off, out_full.data_ptr() + off, stride, 64, T,
                             clean.data_ptr() + off if clean is not None else None)

    torch.cuda.synchronize()
    for i in range(W):
        step(i)
    torch.cuda.synchronize()
    batch.reset_timers()
    adist.barrier(local_rank)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(W + i)
    torch.cuda.synchronize()
    adist.barrier(local_rank)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms_total, launches = batch.timers()
    assert launches == K, (launches, K)

    frames, wall_max, kernel_ms_max = adist.gather_counters(S * T * K, wall, kernel_ms_total,
                                                            device if args.dist_backend == "nccl" else torch.device("cpu"))
    if rank == 0:
        value = frames / wall_max
        kern_avg_s = kernel_ms_total / launches / 1e3
        algo_bytes = ALGO_BYTES_PER_FRAME + (128 if args.clean else 0)
        achieved = algo_bytes * S * T / kern_avg_s / 1e9
        workload_key = f"S{S}_T{T}_fs{args.fs}"
        res = {
            "metric": "AECM frames/sec (64-sample @16kHz) per GPU; bit-exact vs aecm_core_c.cc",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall_max / K * 1e3, "higher_is_better": True, "scaling": "strong" if args.total_streams else "weak",
            "vs_baseline": None, "dtype": "int16/int32 (Q-format fixed point)", "data": "synthetic",
            "config": {"workload": f"{S} streams/GPU x {T} blocks/step, {args.fs} Hz (BASELINE.json configs[2]; "
                                   f"configs[1] = --streams 4096), cng on, echoMode 1{', clean near-end input' if args.clean else ''}, inputs resident in HBM",
                       "streams_per_gpu": S, "blocks_per_step": T, "fs": args.fs, "kernel_variant": args.variant,
                       "sharding": f"static, {world} x {S} independent streams, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": load_traffic(workload_key),
                         "kernel": f"aecm_process_kernel<{args.variant},{'clean' if args.clean else 'noclean'}>",
                         "kernel_avg_ms": kern_avg_s * 1e3, "algorithmic_bytes_per_frame": algo_bytes,
                         "note": "instruction-issue-bound integer kernel (SURVEY.md 8.d): 384 B/frame cannot approach the HBM peak; "
                                 "see issue_bound for the binding resource",
                         "issue_bound": load_issue_note()},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.fs)
        # RCCL writes its version banner to C stdout; flush that first so the JSON is the last line we emit
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res), flush=True)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
