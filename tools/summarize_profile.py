#!/usr/bin/env python3
"""Condense gpurun_out/prof_* (tools/profile_gpu.sh) into profiles/<tag>_*.{csv,json}."""
import collections
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "gpurun_out"
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
S, T, FS = 65536, 128, 16000
frames = S * T


def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "aecm_process" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


rows = list(csv.reader(open(SRC / "prof_stats" / "bench_kernel_stats.csv")))
with open(ROOT / "profiles" / f"{tag}_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    for r in rows[:6]:
        w.writerow([r[0][:110]] + r[1:])
pmc = {}
for p in ("prof_fetch", "prof_write", "prof_sq1", "prof_sq2", "prof_grbm"):
    pmc.update(agg(SRC / p / "bench_counter_collection.csv"))
cal = agg(SRC / "prof_fetch_cal" / "bench_counter_collection.csv")
known_read = frames * 256 + S * (17 * 256 + 256)          # inputs + one state load, no history reads (fixed delay 0)
fetch_factor = known_read / (cal["FETCH_SIZE"] * 1024)
kern_ns = float(rows[1][3])
gui = pmc["GRBM_GUI_ACTIVE"] / 8                          # summed over the 8 XCDs
fetch = pmc["FETCH_SIZE"] * 1024 * fetch_factor
write = pmc["WRITE_SIZE"] * 1024
valu_per_simd = pmc["SQ_INSTS_VALU"] / 1024
summary = {
    "command": "tools/profile_gpu.sh: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 10 "
               "--warmup 2; PMC counters in separate --pmc passes (FETCH_SIZE, WRITE_SIZE, 2 x SQ, GRBM)",
    "workload": {"streams": S, "blocks_per_launch": T, "frames_per_launch": frames, "fs": FS},
    "kernel": rows[1][0][:80],
    "kernel_avg_ms_rocprof": kern_ns / 1e6, "kernel_calls": int(rows[1][1]),
    "frames_per_s_from_kernel_time": frames / (kern_ns / 1e9),
    "pmc_avg_per_launch": pmc,
    "fetch_size_calibration": {"known_read_bytes": known_read, "raw_fetch_bytes": cal["FETCH_SIZE"] * 1024,
                               "factor": fetch_factor,
                               "how": "bench.py --fixed-delay 0: no far-history reads, reads = 256 B/frame + 4608 B/stream"},
    "derived": {
        "algorithmic_bytes_per_launch": 384 * frames,
        "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
        "hbm_GBps": (fetch + write) / (kern_ns / 1e9) / 1e9,
        "valu_insts_per_frame": pmc["SQ_INSTS_VALU"] / frames, "salu_insts_per_frame": pmc["SQ_INSTS_SALU"] / frames,
        "lds_insts_per_frame": pmc["SQ_INSTS_LDS"] / frames,
        "lds_bank_conflict_cycles_per_frame": pmc["SQ_LDS_BANK_CONFLICT"] / frames,
        "shader_cycles_per_launch": gui, "effective_clock_GHz": gui / kern_ns,
        # Issue ports (profiles/r01_issue_port_experiments.md): a SIMD accepts one wave64 VALU and one
        # scalar instruction per 4 shader cycles.  SQ_ACTIVE_INST_VALU counts VALU issue slots of 4 cycles
        # (8-cycle instructions such as v_permlane*_swap / v_sqrt_f32 count twice).
        "ns_per_valu_inst_per_simd": kern_ns * 1024 / pmc["SQ_INSTS_VALU"],
        "valu_port_busy_frac": pmc["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / gui,
        "scalar_port_busy_frac": pmc["SQ_ACTIVE_INST_SCA"] * 4 / 1024 / gui,
        "wave_cycle_split": {k: pmc[k] / pmc["SQ_WAVE_CYCLES"] for k in
                             ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")},
        "avg_resident_waves_per_simd": pmc["SQ_WAVE_CYCLES"] * 4 / 1024 / gui,
    },
}
(ROOT / "profiles" / f"{tag}_rocprof_summary.json").write_text(json.dumps(summary, indent=1))
traffic = {f"S{S}_T{T}_fs{FS}": {"hbm_bytes_per_launch": fetch + write, "read": fetch, "write": write,
                                "source": f"profiles/{tag}_rocprof_summary.json"}}
(ROOT / "profiles" / "hbm_traffic.json").write_text(json.dumps(traffic, indent=1))
print(json.dumps(summary["derived"], indent=1))
print("kernel avg ms", kern_ns / 1e6, "fetch factor", fetch_factor)
