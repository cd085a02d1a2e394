#!/usr/bin/env python3
"""Condense gpurun_out/prof_* (tools/profile_gpu.sh) into profiles/<tag>_*.{csv,json}.

    python tools/summarize_profile.py r02

The workload (streams, blocks per launch, rate) is read from the bench JSON line of the profiled run; the kernel
fingerprint and commit from gpurun_out/prof_meta.json, so that bench.py can tell whether the figures belong to the
library it is timing (webrtc_aecm_amd/isa_census.py)."""
import collections
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "gpurun_out"
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
P = sys.argv[2] if len(sys.argv) > 2 else "prof"         # source directories gpurun_out/<P>_* (tools/profile_gpu.sh: PREFIX)
N_SIMD = 256 * 4


def agg(path, kernel_substr="aecm_process"):
    d = collections.defaultdict(list)
    if not Path(path).exists():
        return {}
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


def bench_line(log):
    for ln in reversed(Path(log).read_text().splitlines()):
        if ln.startswith("{") and '"metric"' in ln:
            return json.loads(ln)
    raise SystemExit(f"no bench JSON line in {log}")


meta = json.loads((SRC / f"{P}_meta.json").read_text())
bl = bench_line(SRC / f"{P}_stats.log")
S, T, FS = bl["config"]["streams_per_gpu"], bl["config"]["blocks_per_step"], bl["config"]["fs"]
frames = S * T
state_bytes = meta["state_size_bytes"] - 32 - 100 * 64 * 2           # vec + scal of one stream (header and history excluded)

rows = list(csv.reader(open(SRC / f"{P}_stats" / "bench_kernel_stats.csv")))
with open(ROOT / "profiles" / f"{tag}_kernel_stats.csv", "w", newline="") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 10 --warmup 2  (commit {meta['commit']}, "
            f"{S} streams x {T} blocks per launch)\n")
    w = csv.writer(f)
    for r in rows[:6]:
        w.writerow([r[0][:110]] + r[1:])
kern_row = next(r for r in rows[1:] if "aecm_process" in r[0])
kern_ns = float(kern_row[3])
pmc = {}
for p in ("fetch", "write", "sq1", "sq2", "sq3", "grbm"):
    pmc.update(agg(SRC / f"{P}_{p}" / "bench_counter_collection.csv"))
cal = agg(SRC / f"{P}_fetch_cal" / "bench_counter_collection.csv")
# chunk-queue launches (bench line: config.launch_chunk_blocks) load and store a stream's state once per chunk
import re as _re
chunk = bl["config"].get("launch_chunk_blocks")
if chunk is None:
    m = _re.search(r"items of (\d+) blocks", bl["roofline"].get("launch_form", ""))
    chunk = int(m.group(1)) if m else 0
state_trips = -(-T // chunk) if chunk else 1
known_read = frames * 256 + S * state_bytes * state_trips  # inputs + the state loads, no history reads (fixed delay 0)
fetch_factor = known_read / (cal["FETCH_SIZE"] * 1024)
gui = pmc["GRBM_GUI_ACTIVE"] / 8                          # summed over the 8 XCDs
fetch = pmc["FETCH_SIZE"] * 1024 * fetch_factor
write = pmc["WRITE_SIZE"] * 1024
valu = pmc["SQ_INSTS_VALU"]
derived = {
    "algorithmic_bytes_per_launch": 384 * frames,
    "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
    "hbm_GBps": (fetch + write) / (kern_ns / 1e9) / 1e9,
    "valu_insts_per_frame": valu / frames, "salu_insts_per_frame": pmc["SQ_INSTS_SALU"] / frames,
    "branch_insts_per_frame": pmc.get("SQ_INSTS_BRANCH", 0) / frames, "smem_insts_per_frame": pmc.get("SQ_INSTS_SMEM", 0) / frames,
    "lds_insts_per_frame": pmc["SQ_INSTS_LDS"] / frames,
    "lds_bank_conflict_cycles_per_frame": pmc["SQ_LDS_BANK_CONFLICT"] / frames,
    "shader_cycles_per_launch": gui, "effective_clock_GHz": gui / kern_ns,
    # VALU issue.  Two denominators: the 4 shader cycles a wave64 instruction of the integer
    # VOP3 / multiply / DPP class occupies the port (measured, profiles/r01_issue_port_experiments.md), and the 2 cycles
    # of the SIMD-32 rate of MI355X_MICROARCH.md that only back-to-back simple VOP2 ops approach.
    "cycles_per_valu_inst": gui * N_SIMD / valu,
    "ns_per_valu_inst_per_simd": kern_ns * N_SIMD / valu,
    "valu_port_busy_frac": pmc["SQ_ACTIVE_INST_VALU"] * 4 / N_SIMD / gui,
    "valu_port_busy_frac_simd32": valu * 2 / N_SIMD / gui,
    "scalar_port_busy_frac": pmc["SQ_ACTIVE_INST_SCA"] * 4 / N_SIMD / gui,
    "thread_cycles_valu_per_inst": pmc.get("SQ_THREAD_CYCLES_VALU", 0) / valu if valu else None,
    "wave_cycle_split": {k: pmc[k] / pmc["SQ_WAVE_CYCLES"] for k in
                         ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")},
    "avg_resident_waves_per_simd": pmc["SQ_WAVE_CYCLES"] * 4 / N_SIMD / gui,
}
workload_key = f"S{S}_T{T}_fs{FS}"
summary = {
    "command": "tools/profile_gpu.sh: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 10 "
               "--warmup 2; PMC counters in separate --pmc passes (FETCH_SIZE, WRITE_SIZE, 3 x SQ, GRBM), --kernel-trace only",
    "measured_at_commit": meta["commit"], "kernel_symbol": meta.get("kernel_symbol"), "kernel_fingerprint": meta["kernel_fingerprint"],
    "static_counts": meta["static_counts"], "static_valu_fast_class": meta["static_valu_fast_class"],
    "workload": {"streams": S, "blocks_per_launch": T, "frames_per_launch": frames, "fs": FS, "launch_chunk_blocks": chunk,
                 "state_round_trips_per_launch": state_trips},
    "kernel": kern_row[0][:80],
    "kernel_avg_ms_rocprof": kern_ns / 1e6, "kernel_calls": int(kern_row[1]),
    "kernel_avg_ms_hip_events_same_run": bl["roofline"]["kernel_avg_ms"],
    "frames_per_s_from_kernel_time": frames / (kern_ns / 1e9),
    "pmc_avg_per_launch": pmc,
    "fetch_size_calibration": {"known_read_bytes": known_read, "raw_fetch_bytes": cal["FETCH_SIZE"] * 1024,
                               "factor": fetch_factor,
                               "how": f"bench.py --fixed-delay 0: no far-history reads, reads = 256 B/frame + {state_bytes} B/stream x {state_trips} state load(s) per launch"},
    "derived": derived,
    "traffic_by_workload": {workload_key: fetch + write},
}
# the streaming path (tools/bench_sessions.py): second roofline entry, HBM-side
tick_stats = SRC / f"{P}_tick" / "tick_kernel_stats.csv"
if tick_stats.exists():
    trows = list(csv.reader(open(tick_stats)))
    with open(ROOT / "profiles" / f"{tag}_tick_kernel_stats.csv", "w", newline="") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python tools/bench_sessions.py --streams 65536 --ticks 200  (commit {meta['commit']})\n")
        w = csv.writer(f)
        for r in trows[:6]:
            w.writerow([r[0][:110]] + r[1:])
    tick = {"kernels": {}}
    per_tick_ns = 0.0
    n_ticks = 240
    for r in trows[1:]:
        if "aecm" not in r[0]:
            continue
        name = r[0].split("(")[0].split("::")[-1][:40]
        tick["kernels"][name] = {"calls": int(r[1]), "avg_us": float(r[3]) / 1e3}
        per_tick_ns += float(r[2]) / n_ticks
    tf = {}
    for sub in ("aecm_process", "aecm_tick"):
        for d in ("tick_fetch", "tick_write"):
            for k, v in agg(SRC / f"{P}_{d}" / "tick_counter_collection.csv", sub).items():
                tf[f"{sub}:{k}"] = v
    tick["pmc_avg_per_launch"] = tf
    # the kernel that runs the tick's blocks
    for sub, pat in (("aecm_tick", "aecm_tick_flow"),):
        if f"{sub}:FETCH_SIZE" in tf and f"{sub}:WRITE_SIZE" in tf and any(pat in r[0] for r in trows[1:]):
            rd = tf[f"{sub}:FETCH_SIZE"] * 1024 * fetch_factor
            wr = tf[f"{sub}:WRITE_SIZE"] * 1024
            avg_ns = next(float(r[3]) for r in trows[1:] if pat in r[0])
            tick["block_kernel"] = {"kernel": pat, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "avg_ms": avg_ns / 1e6,
                                    "hbm_GBps": (rd + wr) / (avg_ns / 1e9) / 1e9, "hbm_frac_of_8TBps": (rd + wr) / (avg_ns / 1e9) / 8e12,
                                    "bytes_per_session_tick": (rd + wr) / 65536}
            break
    sq = agg(SRC / f"{P}_tick_sq" / "tick_counter_collection.csv", "aecm_tick")
    if sq:
        # instruction issue of the tick kernel per session-tick (2.5 blocks per 16 kHz tick): what a tick costs on top of
        # its blocks (state unpack / pack, table fill, ring appends, output assembly)
        tick["insts_per_session_tick"] = {k: v / 65536 for k, v in sq.items()}
    tick["gpu_ms_per_tick_sum_of_kernels"] = per_tick_ns / 1e6
    try:
        tick["bench_sessions_line"] = json.loads([ln for ln in (SRC / f"{P}_tick.log").read_text().splitlines() if ln.startswith("{")][-1])
    except Exception:
        pass
    summary["streaming_tick"] = tick
(ROOT / "profiles" / f"{tag}_rocprof_summary.json").write_text(json.dumps(summary, indent=1))
print(json.dumps(summary["derived"], indent=1))
print("kernel avg ms", kern_ns / 1e6, "fetch factor", fetch_factor)
if "streaming_tick" in summary:
    print(json.dumps(summary["streaming_tick"], indent=1)[:1500])
