#!/usr/bin/env python3
"""Headline benchmark: AECM 64-sample frames/s (WebRtcAecm_ProcessBlock-equivalents) on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input: a single launch that advances
every one of the S resident streams by T blocks (S*T frames).  Inputs are synthetic far/near pairs
generated on the device and resident in HBM before the timed region; state stays on the device between
steps.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--blocks T] [--fs 16000|8000]

N > 1: one rank per GPU.  Started by hand (`python bench.py --gpus N`, no WORLD_SIZE in the environment) the
script re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N`; started by
torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE.  Every rank owns S streams (weak scaling, no
data-path collective); RCCL only lines the ranks up and gathers the counters.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# dmabuf IPC only on this driver: RCCL needs it, and the HSA runtime reads it when torch first touches the GPU -- so before
# anything imports torch (the driver's environment already exports it; this covers a bare shell)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ALGO_BYTES_PER_FRAME = 384            # 128 far in + 128 near in + 128 out
HBM_PEAK_GBPS = 8000.0                # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# rocprofv3 records of this round's kernels (tools/gpu_round_final.sh): the headline (chunk-queue kernel), the configs[1] kernel (pipelined,
# balanced) and the small launches' kernel (pipelined, sixteen waves per workgroup)
PROFILE_SUMMARIES = [ROOT / "profiles" / "r06_rocprof_summary.json", ROOT / "profiles" / "r06_pipelined_rocprof_summary.json",
                     ROOT / "profiles" / "r06_small_rocprof_summary.json"]


PROFILES = {
    # name: (far-end envelope levels, near-end talk levels, envelope segment in samples, what it is)
    "recipe": ([15., 60., 500., 3000., 9000., 20000.], [0., 0., 0., 2000., 8000.], 6400,
               "the default: far end with pauses (0.4 s segments from 15 to 20 000), echo + near-end talk bursts 2 segments in 5"),
    "always_active": ([500., 3000., 9000., 20000.], [2000., 8000.], 6400,
                      "far end never below 500 and near-end talk in every segment"),
    "double_talk": ([2000., 30000.], [60000.], 128,
                    "far end switching between 2 000 and 30 000 every 8 ms under full-scale near-end talk: the VAD fires on the loud half, the "
                    "suppression gain never decays to zero (no pass-through blocks), the output is loud enough to rescale inverse-transform stages"),
    "full_scale": ([60000.], [60000.], 6400,
                   "both ends clipped to +-32767 nearly everywhere: every inverse-transform stage rescales -- but a far end without level "
                   "changes never trips the VAD (farEnergyMaxMin stays 0, aecm_core.cc:733-740), so no NLMS and gain zero: a CHEAP case"),
    "silent": ([0.], [0.], 6400, "digital silence on both ends"),
}


def synth_on_device(torch, S, L, seed, device, chunk=8192, profile="recipe"):
    """Synthetic far/near int16 [S, L] in HBM: white noise x piecewise-constant envelope (0.4 s
    segments from the profile's levels), near = sparse 4-tap echo of far + near-end talk bursts (the recipe
    of webrtc_aecm_amd/synth.py, float math on the GPU; bench data need not be reproducible bit
    for bit across machines, parity is checked elsewhere).  profile: PROFILES (the content sweep)."""
    far = torch.empty((S, L), dtype=torch.int16, device=device)
    near = torch.empty((S, L), dtype=torch.int16, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    levels = torch.tensor(PROFILES[profile][0], device=device)
    talk_levels = torch.tensor(PROFILES[profile][1], device=device)
    seg = PROFILES[profile][2]
    nseg = L // seg + 2
    taps = ((100, 0.5), (180, -0.3), (333, 0.2), (600, 0.1))
    for s0 in range(0, S, chunk):
        n = min(chunk, S - s0)
        env = levels[torch.randint(0, len(levels), (n, nseg), generator=g, device=device)].repeat_interleave(seg, dim=1)[:, :L]
        x = torch.randn((n, L), generator=g, device=device) * env * 0.58
        x[:, 1:-1] = (x[:, :-2] + 2 * x[:, 1:-1] + x[:, 2:]) * 0.25
        x = x.clamp_(-32768, 32767).round_()
        echo = torch.zeros_like(x)
        for d, gain in taps:
            echo[:, d:] += gain * x[:, :-d]
        tenv = talk_levels[torch.randint(0, len(talk_levels), (n, nseg), generator=g, device=device)].repeat_interleave(seg, dim=1)[:, :L]
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


def cpu_baseline(fs, pairs, budget_s=12.0):
    """The reference C path (oracle/_ref, kind 'reference') or our restatement of it (kind 'port') timed
    on this box's host cores: one stream per thread, bounded to ~budget_s of wall time.  `pairs` are
    (far, near) int16 host copies of streams of the GPU leg's own workload (the same inputs, same config)."""
    import threading

    from oracle import pyoracle
    cores = usable_cores()
    use_ref = pyoracle.have_reference()
    mk = (lambda: pyoracle.RefCoreStream(fs, 1, 1)) if use_ref else (lambda: pyoracle.OracleStream(fs, 1, 1))
    chunk = 4096 * 64                              # samples per call (~40 ms of CPU work)
    pairs = [(np.ascontiguousarray(f[:chunk]), np.ascontiguousarray(d[:chunk])) for f, d in pairs]
    blocks_per_call = pairs[0][0].size // 64
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
            done += blocks_per_call
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
        "sample": f"{cores} threads (one stream each) x {dt:.1f} s wall = {total} frames; inputs = streams 0..{len(pairs) - 1} of the "
                  f"GPU leg's own batch ({fs} Hz, cng on, echoMode 1), {blocks_per_call} blocks per call; "
                  + ("unmodified reference built -O2 from /root/reference (oracle/_ref)" if use_ref
                     else "oracle/aecm_oracle.c restatement built -O2"),
    }


def verify_timed_workload(batch, far, near, clean, out, S, T, passes, fs, fixed_delay, timed_passes=0, world=1):
    """Parity of the timed workload itself (outside the timed region): streams 0..15, S/2 and S-1 of this rank's own
    batch are pushed through the CPU checker -- the unmodified reference (oracle/_ref, WebRtcAecm_ProcessBlock,
    aecm_core_c.cc:368-711) when its prebuilt library travelled with the tree, our restatement otherwise -- for every
    block the GPU processed (warm-up + timed passes over the T-block input), and the checker's output of the LAST pass
    and its 24-word state digest must equal the GPU's `out` rows and WebRtcAecmBatch_GetDigest bit for bit."""
    from oracle import pyoracle
    use_ref = pyoracle.have_reference()
    picks = sorted(set(list(range(min(16, S))) + [S // 2, S - 1]))
    host = [(far[i].cpu().numpy(), near[i].cpu().numpy(), None if clean is None else clean[i].cpu().numpy(),
             out[i].cpu().numpy()) for i in picks]
    gpu_digest = [batch.digest(i) for i in picks]

    def one(k):
        f, d, c, got = host[k]
        chk = pyoracle.RefCoreStream(fs, 1, 1) if use_ref else pyoracle.OracleStream(fs, 1, 1)
        # which paths the blocks of the TIMED passes took: counted by the restatement (oracle/aecm_oracle.h: ORC_STAT_*), which runs
        # next to the reference when that is the checker (the reference has no such counters and is not modified)
        cnt = chk if not use_ref else (pyoracle.OracleStream(fs, 1, 1) if c is None else None)
        if fixed_delay >= 0:
            chk.control(fixed_delay, 1)
            if cnt is not None and cnt is not chk:
                cnt.control(fixed_delay, 1)
        exp = None
        before = None
        for p in range(passes):
            if p == passes - timed_passes and cnt is not None:
                before = cnt.stats()
            if c is None:
                exp = chk.process(f, d)
                if cnt is not None and cnt is not chk:
                    cnt.process(f, d)
            else:
                exp = np.concatenate([chk.process_block_clean(f[b * 64:(b + 1) * 64], d[b * 64:(b + 1) * 64], c[b * 64:(b + 1) * 64])
                                      for b in range(T)])
        bad_out = int(np.count_nonzero(exp != got))
        dig_ok = bool(np.array_equal(chk.digest(), gpu_digest[k]))
        stats = None
        if cnt is not None and before is not None:
            after = cnt.stats()
            stats = {n: after[n] - before[n] for n in after}
        return bad_out, dig_ok, stats
    t0 = time.perf_counter()
    # every rank checks its own shard at the same time: a rank takes its share of the host's cores, not all of them
    workers = max(1, min(len(picks), usable_cores() // max(1, world)))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        res = list(ex.map(one, range(len(picks))))
    bad = [picks[k] for k, (b, dg, _) in enumerate(res) if b or not dg]
    tot = {}
    for _, _, st in res:
        for n, v in (st or {}).items():
            tot[n] = tot.get(n, 0) + v
    shares = {n + "_share": round(v / tot["blocks"], 4) for n, v in tot.items() if n != "blocks"} if tot.get("blocks") else None
    return {"streams": picks, "blocks": passes * T, "blocks_compared_sample_by_sample": T, "content_shares": shares,
            "checker": "reference" if use_ref else "port", "checker_threads": workers, "ok": not bad, "mismatching_streams": bad,
            "state_digest_compared": True, "seconds": time.perf_counter() - t0,
            "what": f"{len(picks)} streams of the timed batch x all {passes} passes ({passes * T} blocks each) re-run on the CPU "
                    f"checker; last pass's output rows and the final state digest compared bit for bit"}


def workload_name(S, T, fs, world, clean):
    """Which BASELINE.json configuration this run is (by streams per GPU, rate and GPU count)."""
    if clean:
        tag = "custom (clean near-end input: third transform per block)"
    elif fs == 16000 and S == 65536 and world == 8:
        tag = "BASELINE.json configs[4] (524288 streams over 8 GPUs)"
    elif fs == 16000 and S == 65536:
        tag = "BASELINE.json configs[2] (65536 streams, 16 kHz)" + (f" on each of {world} GPUs" if world > 1 else "")
    elif fs == 16000 and S == 4096:
        tag = "BASELINE.json configs[1] (4096 streams, 16 kHz)"
    elif fs == 8000 and S == 32768:
        tag = "BASELINE.json configs[3] (8 kHz mode, 32768 streams)"
    else:
        tag = "custom size"
    return f"{tag}: {S} streams/GPU x {T} blocks/step, {fs} Hz, cng on, echoMode 1, inputs resident in HBM"   # (+ the content profile, main())


def load_profile_record(lib_path, workload_key, kernel_substr):
    """Issue-port figures and HBM traffic of the dominant kernel come from separate rocprofv3 PMC passes
    (tools/profile_gpu.sh -> tools/summarize_profile.py -> profiles/r06_*rocprof_summary.json), which cannot run
    inside this process.  They are only valid for the binary they were measured on: the summary stores the
    instruction-stream fingerprint of the profiled kernel (webrtc_aecm_amd/isa_census.py) and is quoted only
    when the library timed here has the same fingerprint and the same workload; otherwise it is reported as stale."""
    from webrtc_aecm_amd import isa_census
    try:
        now = isa_census.census(lib_path, kernel_substr)
    except Exception as e:
        return None, {"available": False, "reason": f"could not disassemble {lib_path}: {e}"}
    static = {"kernel_symbol": now["kernel"], "kernel_fingerprint": now["fingerprint"], "static_counts": now["counts"],
              "static_valu_fast_class": now["valu_fast_class"], "static_valu_8cycle_class": now["valu_8cycle_class"]}
    recs = [(p, json.loads(p.read_text())) for p in PROFILE_SUMMARIES if p.exists()]
    if not recs:
        return None, dict(static, available=False, reason=f"{PROFILE_SUMMARIES[0].name} not recorded yet")
    same_kernel = [(p, r) for p, r in recs if r.get("kernel_symbol") == now["kernel"]]
    match = [(p, r) for p, r in same_kernel if r.get("kernel_fingerprint") == now["fingerprint"]]
    if not match:
        p, r = (same_kernel or recs)[0]
        return None, dict(static, available=False, stale=bool(same_kernel),
                          reason=(f"profiles/{p.name} was measured on kernel fingerprint {r.get('kernel_fingerprint')} (commit {r.get('measured_at_commit')}), "
                                  f"this library is {now['fingerprint']}: re-profile") if same_kernel else
                                 f"no rocprofv3 record of {now['kernel'][:60]} under profiles/ (recorded: the headline, the configs[1] and the sixteen-wave kernels)")
    PROFILE_SUMMARY, rec = match[0]
    d = rec.get("derived", {})
    note = dict(static, available=True, measured_at_commit=rec.get("measured_at_commit"),
                valu_insts_per_frame=d.get("valu_insts_per_frame"), salu_insts_per_frame=d.get("salu_insts_per_frame"),
                branch_insts_per_frame=d.get("branch_insts_per_frame"),
                cycles_per_valu_inst=d.get("cycles_per_valu_inst"),
                valu_port_busy_frac_at_4_cycles_per_inst=d.get("valu_port_busy_frac"),
                valu_port_busy_frac_at_2_cycles_per_inst=d.get("valu_port_busy_frac_simd32"),
                scalar_port_busy_frac=d.get("scalar_port_busy_frac"),
                model="denominators: one wave64 VALU instruction per SIMD per 4 shader cycles (measured for the integer VOP3 / "
                      "multiply / DPP class this kernel is mostly made of) resp. per 2 cycles (the SIMD-32 rate of "
                      "MI355X_MICROARCH.md, reached only by back-to-back simple VOP2 ops)",
                source=f"profiles/{PROFILE_SUMMARY.name} (rocprofv3 SQ_* PMC passes), profiles/r02_valu_class_census.md")
    traffic = rec.get("traffic_by_workload", {}).get(workload_key)
    return traffic, note


def live_counters(child_args, kernel_regex, frames_per_launch, kernel_avg_s, compute_units):
    """HBM traffic and issued instructions of the dominant kernel measured IN THIS RUN (not quoted from profiles/): three short
    rocprofv3 --pmc passes over child runs of this script with the same workload (counters in their own passes with --kernel-trace
    only; FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, rocprofv3 section), averaged per launch of the kernel
    this run timed.  Corrections as that guide prescribes for gfx950: the counters are in KiB; FETCH_SIZE tallies the kernel's wide
    coalesced reads at half their bytes (x 2: profiles/r06_rocprof_summary.json calibrates 1.998 on a known byte count of this very
    access pattern); WRITE_SIZE as it is.  None (with the reason) where rocprofv3 is missing or a pass fails."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if Path("/opt/rocm/bin/rocprofv3").exists() else None)
    if exe is None:
        return {"available": False, "reason": "rocprofv3 not found"}
    out = {}
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.perf_counter()
    for counters in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES", "GRBM_GUI_ACTIVE"]):
        d = tempfile.mkdtemp(prefix="aecm_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "b", "--",
                   sys.executable, str(Path(__file__).resolve()), *child_args, "--no-cpu-baseline", "--no-parity", "--no-live-counters", "--steps", "2", "--warmup", "1"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            vals = {}
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if re.search(kernel_regex, row["Kernel_Name"]) or kernel_regex.split("IL")[0].replace("aecm_", "aecm::aecm_") in row["Kernel_Name"]:
                        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            if r.returncode != 0 or not vals:
                return {"available": False, "reason": f"rocprofv3 --pmc {' '.join(counters)} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}"}
            out.update({k: sum(v) / len(v) for k, v in vals.items()})
        except (OSError, subprocess.SubprocessError) as e:
            return {"available": False, "reason": f"rocprofv3 pass failed: {e}"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd, wr = out["FETCH_SIZE"] * 1024 * 2.0, out["WRITE_SIZE"] * 1024
    # issue: shader cycles of the launch (GRBM_GUI_ACTIVE is summed over the 8 XCDs) x SIMDs / wave64 VALU instructions
    cycles = out.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    per_valu = cycles * compute_units * 4 / out["SQ_INSTS_VALU"] if cycles and out.get("SQ_INSTS_VALU") else None
    return {"available": True, "how": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE) over child runs of this "
                                      "command, 3 launches each, averaged per launch of the timed kernel; FETCH_SIZE x 1024 x 2 (gfx950), WRITE_SIZE x 1024",
            "shader_cycles_per_launch": cycles or None, "cycles_per_valu_inst": per_valu,
            "valu_port_busy_frac_at_4_cycles_per_inst": 4.0 / per_valu if per_valu else None,
            "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
            "hbm_GBps_at_this_runs_kernel_time": (rd + wr) / kernel_avg_s / 1e9,
            "valu_insts_per_frame": out["SQ_INSTS_VALU"] / frames_per_launch, "salu_insts_per_frame": out["SQ_INSTS_SALU"] / frames_per_launch,
            "waves_per_launch": out.get("SQ_WAVES"), "seconds": time.perf_counter() - t0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=65536, help="streams per GPU (BASELINE config 3: 65536)")
    ap.add_argument("--blocks", type=int, default=1280,
                    help="blocks per stream per step (one launch); 1280 makes a 20-step timed region >= 2 s at 65536 streams")
    ap.add_argument("--total-streams", type=int, default=0,
                    help="strong scaling: this many streams in total, split evenly over the GPUs (overrides --streams)")
    ap.add_argument("--clean", action="store_true",
                    help="also feed a clean near-end input (WebRtcAecm_ProcessBlock's nearendClean: third transform per block)")
    ap.add_argument("--fs", type=int, default=16000)
    ap.add_argument("--variant", choices=["fast", "safe"], default="fast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the CPU re-run of the timed workload (A/B timing loops only)")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="skip the rocprofv3 counter passes that measure roofline.traffic in this very run (N = 1 only; ~40 s)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for the counter gather (nccl = RCCL)")
    ap.add_argument("--share-devices", action="store_true",
                    help="map rank r to HIP device r %% device_count (exercises the N-rank path on a box with fewer GPUs; "
                         "implies --dist-backend gloo, RCCL refuses two ranks on one device)")
    ap.add_argument("--launch-blocks", type=int, default=0,
                    help="experiment: split every step into launches of this many blocks over the same input (0 = one launch per "
                         "step, the measured configuration); separates launch-length effects from the data's")
    ap.add_argument("--profile", choices=sorted(PROFILES), default="recipe",
                    help="content of the synthetic signals (the frame rate depends on which data-dependent paths the blocks take): "
                         + "; ".join(f"{k} = {v[3]}" for k, v in PROFILES.items()))
    ap.add_argument("--policy", default="",
                    help="launch-policy fields to change, 'field=value field=value' of AecmLaunchPolicy (include/aecm_batch.h), e.g. "
                         "'queue_chunk_blocks=0' (one wavefront per stream always) or 'pipelined_min_streams=0'; A/B runs -- the default is the shipped policy")
    ap.add_argument("--fixed-delay", type=int, default=-1,
                    help="WebRtcAecm_Control fixed delay (>= 0 disables the estimator's choice; 0 = no far-history reads; "
                         "used to calibrate the FETCH_SIZE counter on a known byte count)")
    args = ap.parse_args()

    from webrtc_aecm_amd import dist as adist
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import torch
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < args.gpus and not args.share_devices:      # one clear line, before any rendezvous
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {n_dev} HIP device(s) visible on this node "
                             f"(--share-devices maps ranks onto the devices present)")
        adist.self_launch(str(Path(__file__).resolve()), sys.argv[1:], args.gpus)      # does not return

    import torch

    import webrtc_aecm_amd as aecm

    if args.share_devices:
        args.dist_backend = "gloo"
    # everything that can be wrong with the launch is checked BEFORE the rendezvous, so a bad launch dies in seconds with
    # one line per rank instead of hanging in the process-group set-up
    rank, local_rank, world = adist.env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the AECM hot path has no CPU implementation in the product")
    n_dev = torch.cuda.device_count()
    if args.share_devices:
        local_rank %= n_dev
    elif local_rank >= n_dev:
        raise SystemExit(f"bench.py: rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} HIP device(s) visible "
                         f"(--share-devices maps ranks onto the devices present)")
    torch.cuda.set_device(local_rank)       # before init_process_group: RCCL binds the communicator to the current device
    adist.init(args.dist_backend, local_rank)
    device = torch.device("cuda", local_rank)

    S, T, K, W = args.streams, args.blocks, args.steps, args.warmup
    if args.total_streams:
        _, S = adist.shard_range(args.total_streams, rank, world)
    far, near = synth_on_device(torch, S, T * 64, 1234 + rank, device, profile=args.profile)
    clean = (near.to(torch.int32) * 3 // 4).to(torch.int16) if args.clean else None
    batch = aecm.AecmBatch(S, args.fs, cng_mode=1, echo_mode=1, device=local_rank,
                           variant=aecm.KERNEL_FAST if args.variant == "fast" else aecm.KERNEL_SAFE)
    if args.policy:
        batch.set_launch_policy(**{kv.split("=", 1)[0]: int(kv.split("=", 1)[1], 0) for kv in args.policy.split()})
    if args.fixed_delay >= 0:
        batch.control(args.fixed_delay, 1)
    stride = far.shape[1]
    out = torch.empty_like(near)            # same [S][T*64] layout as the inputs

    C = args.launch_blocks if 0 < args.launch_blocks < T else T
    chunks = [(b0, min(C, T - b0)) for b0 in range(0, T, C)]

    def step():                             # one C-ABI call = one launch = S*T frames; the input repeats every T blocks
        for b0, nb in chunks:               # (one chunk unless --launch-blocks)
            o = b0 * 128                    # bytes into every stream's row
            batch.process_device(far.data_ptr() + o, near.data_ptr() + o, out.data_ptr() + o, stride, 64, nb,
                                 clean.data_ptr() + o if clean is not None else None)

    torch.cuda.synchronize()
    for _ in range(W):
        step()
    batch.synchronize()
    torch.cuda.synchronize()
    batch.reset_timers()
    adist.barrier(local_rank)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    batch.synchronize()                     # the engine runs on its own (non-blocking) HIP stream
    torch.cuda.synchronize()
    adist.barrier(local_rank)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms_total, launches = batch.timers()      # HIP events recorded by the library around every launch, on its stream
    assert launches == K * len(chunks), (launches, K, len(chunks))
    launches = K                            # per step from here on (a step is len(chunks) launches only under --launch-blocks)

    cdev = device if args.dist_backend == "nccl" else torch.device("cpu")
    _form, _detail = batch.describe_launch(C, bool(args.clean))
    dev_name = "%s %d CUs hip%d pci %s form %d/%#x" % (aecm.device_info(local_rank)[0].split(":")[0], aecm.device_info(local_rank)[1], local_rank,
                                                       aecm.device_pci_bus_id(local_rank), _form, _detail)      # (64 bytes travel per rank)
    c = adist.gather_counters(S * T * K, wall, kernel_ms_total, cdev, dev_name)
    parity = None
    if not args.no_parity:                  # every rank checks streams of its own shard; rank 0 reports, all must agree
        parity = verify_timed_workload(batch, far, near, clean, out, S, T, W + K, args.fs, args.fixed_delay, timed_passes=K, world=world)
        parity["ranks_ok"] = adist.all_ok(parity["ok"], cdev)
    if rank == 0:
        value = c["frames"] / c["seconds"]
        kern_avg_s = kernel_ms_total / launches / 1e3
        algo_bytes = ALGO_BYTES_PER_FRAME + (128 if args.clean else 0)
        achieved = algo_bytes * S * T / kern_avg_s / 1e9
        workload_key = f"S{S}_T{T}_fs{args.fs}" + ("_clean" if args.clean else "")
        from webrtc_aecm_amd import isa_census
        form, chunk = batch.describe_launch(C, bool(args.clean))              # which block kernel a launch of this shape takes
        if args.variant == "fast":
            kernel_substr, kernel_name = isa_census.block_kernel(form, bool(args.clean), chunk)
            traffic, issue = load_profile_record(aecm.library_path(), workload_key, kernel_substr)
        else:
            kernel_name, traffic, issue = f"aecm_process_kernel<safe,{'clean' if args.clean else 'noclean'}>", None, None
        if args.clean or len(chunks) > 1:                      # the profile record is of the default workload's launch
            traffic = None
        from webrtc_aecm_amd import build as _b
        bi = _b.build_info()                       # written next to the library when it was built (no .git on the GPU box)
        commit = (bi.get("commit") or "unknown") + ("+dirty" if bi.get("dirty") else "")
        res = {
            "metric": "AECM frames/sec (64-sample @16kHz) per GPU; bit-exact vs aecm_core_c.cc",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": c["seconds"] / K * 1e3, "higher_is_better": True, "scaling": "strong" if args.total_streams else "weak",
            "vs_baseline": None, "dtype": "int16/int32 (Q-format fixed point)", "data": "synthetic",
            "config": {"workload": workload_name(S, T, args.fs, world, args.clean) + ("" if args.profile == "recipe" else f", content profile {args.profile}"),
                       "content_profile": args.profile,
                       "streams_per_gpu": S, "blocks_per_step": T, "launches_per_step": len(chunks), "launch_form": form,
                       "launch_chunk_blocks": chunk if form == 2 else 0, "pipelined_tail_waves": (chunk & 0xff) if form == 3 else None, "fs": args.fs, "kernel_variant": args.variant,
                       "sharding": f"static, {world} x {S} independent streams, no data-path collective",
                       "timed_region_s": c["seconds"], "commit": commit, **({"launch_policy_changes": args.policy} if args.policy else {})},
            "device": dict(zip(("name", "compute_units", "clock_khz"), aecm.device_info(local_rank))),
            "ranks": {"world_size": world, "ranks_seen": c["ranks_seen"], "collective_backend": c["backend"],
                      "devices_visible_to_rank0": n_dev, "share_devices": bool(args.share_devices),
                      "per_rank_streams": [p[0] // (T * K) for p in c["per_rank"]], "per_rank_device": c["names"],
                      "per_rank_frames_per_s": [p[0] / p[1] for p in c["per_rank"]],
                      "per_rank_kernel_ms_per_step": [p[2] / K for p in c["per_rank"]]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": kernel_name,
                         "launch_form": {0: "one wavefront per stream (launch resident at once)", 1: "one wavefront per stream",
                                         2: f"chunk queue: items of {chunk} blocks claimed in order by resident wavefronts",
                                         3: f"pipelined: {4 + (4 if chunk & 0x200 else 2) + (chunk & 0xff) + ((2 if chunk & 0x1000 else 4) if chunk & 0x800 else 0) + (4 if chunk & 0x1000 else 0)} wavefronts per four streams, the forward transforms one block "
                                            f"ahead in {4 if chunk & 0x200 else 2} wavefronts of their own"
                                            + (f", the inverse transforms one block behind in {chunk & 0xff} more" if chunk & 0xff else "")
                                            + (f", the delay estimator one block ahead in {2 if chunk & 0x1000 else 4} more" if chunk & 0x800 else "")
                                            + (", the gain half of the block one block behind the channel half in 4 more" if chunk & 0x1000 else "")
                                            + (", spectra formed by the back wavefronts" if chunk & 0x400 else "")
                                            + (", front-wave priorities balanced by progress feedback" if chunk & 0x100 else "")}[form],
                         "kernel_avg_ms": kern_avg_s * 1e3, "algorithmic_bytes_per_frame": algo_bytes,
                         "algorithmic_bytes_per_launch": algo_bytes * S * T,
                         "note": "instruction-issue-bound integer kernel (SURVEY.md 8.d): 384 B/frame cannot approach the HBM peak; "
                                 "see issue_bound for the binding resource",
                         "issue_bound": issue},
        }
        # roofline.traffic measured in THIS run (N = 1): counter passes over child runs of the same command; the quoted profile record
        # stays beside it (issue_bound: the port-busy fractions need the SQ cycle counters of the full profile)
        if world == 1 and not args.no_live_counters and not args.no_cpu_baseline and args.variant == "fast":
            child = ["--streams", str(S), "--blocks", str(T), "--fs", str(args.fs), "--profile", args.profile]
            if args.clean:
                child.append("--clean")
            if args.launch_blocks:
                child += ["--launch-blocks", str(args.launch_blocks)]
            if args.policy:
                child += ["--policy", args.policy]
            if args.fixed_delay >= 0:
                child += ["--fixed-delay", str(args.fixed_delay)]
            live = live_counters(child, kernel_substr, S * C, kern_avg_s / len(chunks), aecm.device_info(local_rank)[1])
            res["roofline"]["live_counters"] = live
            if live.get("available"):
                res["roofline"]["traffic_quoted_from_profile"] = res["roofline"]["traffic"]
                res["roofline"]["traffic"] = live["hbm_bytes_per_launch"]
                res["roofline"]["traffic_source"] = "measured in this run (live_counters)"
        shares = parity.pop("content_shares", None) if parity is not None else None
        # what the frame rate was measured ON: the signal profile and, from the checker's streams over the timed passes, how many
        # blocks took the data-dependent paths that cost or save work (profiles/r06_content_sweep.txt has all profiles side by side)
        res["content"] = {"profile": args.profile, "what": PROFILES[args.profile][3],
                          "nlms_share": shares and shares.get("nlms_share"), "passthrough_share": shares and shares.get("gain_zero_share"),
                          "q_steady_share": shares and shares.get("q_steady_share"), "ifft_unscaled_share": shares and shares.get("ifft_unscaled_share"),
                          "delayed_share": shares and shares.get("delayed_share"),
                          "measured_on": None if shares is None else f"{len(parity['streams'])} streams of the timed batch x the {K} timed passes"}
        if parity is not None:
            res["parity"] = parity
        if world == 1 and not args.no_cpu_baseline:
            n_pairs = min(usable_cores(), S, 32)
            pairs = [(far[i].cpu().numpy(), near[i].cpu().numpy()) for i in range(n_pairs)]
            res["cpu_baseline"] = cpu_baseline(args.fs, pairs)
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
    if parity is not None and not (parity["ok"] and parity["ranks_ok"] == world):
        raise SystemExit(f"rank {rank}: the timed workload is NOT bit-exact vs the {parity['checker']} "
                         f"(streams {parity['mismatching_streams']}; ranks ok {parity['ranks_ok']}/{world})")


if __name__ == "__main__":
    main()
