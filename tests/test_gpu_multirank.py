"""N-rank path on real hardware without needing N GPUs: two ranks (torch.distributed.run, gloo for the counter
collective) share the one MI355X of the GPU box, each creating its own HIP engine with AecmBatch(device=...) and
owning its static shard of the streams -- the code path `bench.py --gpus N` runs, with RCCL swapped for gloo only
because RCCL refuses two ranks on one device."""
import json
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import numpy as np
import pytest

from helpers import oracle_batch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_two_ranks_shard_8192_streams_on_the_hip_library(tmp_path):
    """Rank r processes streams shard_range(8192, r, 2) through the C ABI on the GPU; the union must equal the
    oracle for every stream (64 distinct seeds replicated), and the gathered counters must prove two ranks took part."""
    total, T, fs, U = 8192, 96, 16000, 64
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {str(ROOT)!r}); sys.path.insert(0, {str(ROOT / 'tests')!r})
        import numpy as np, torch
        import webrtc_aecm_amd as aecm
        from webrtc_aecm_amd import dist as adist
        from helpers import synth_streams
        rank, local_rank, world = adist.init("gloo")
        dev = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev)
        first, count = adist.shard_range({total}, rank, world)
        seeds = [3000 + (s % {U}) for s in range(first, first + count)]
        far, near = synth_streams(seeds[:{U}] if count >= {U} else seeds, {T}, {fs})
        idx = np.arange(count) % far.shape[0]
        dfar = torch.from_numpy(far[idx]).cuda(); dnear = torch.from_numpy(near[idx]).cuda(); dout = torch.empty_like(dnear)
        torch.cuda.synchronize()
        b = aecm.AecmBatch(count, {fs}, device=dev)
        adist.barrier()
        t0 = time.perf_counter()
        b.process_device(dfar.data_ptr(), dnear.data_ptr(), dout.data_ptr(), {T} * 64, 64, {T})
        b.synchronize()
        adist.barrier()
        kms, n = b.timers()
        c = adist.gather_counters(count * {T}, time.perf_counter() - t0, kms, torch.device("cpu"))
        np.save({str(tmp_path)!r} + f"/out_{{rank}}.npy", dout.cpu().numpy())
        np.save({str(tmp_path)!r} + f"/first_{{rank}}.npy", np.array([first, count, dev]))
        if rank == 0:
            import json
            json.dump(dict(frames=c["frames"], ranks_seen=c["ranks_seen"], per_rank=c["per_rank"]), open({str(tmp_path)!r} + "/counters.json", "w"))
        import torch.distributed as dist
        dist.destroy_process_group()
    """))
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", "29631", str(script)], env=_clean_env(), timeout=600)
    c = json.load(open(tmp_path / "counters.json"))
    assert c["ranks_seen"] == 2 and c["frames"] == total * T and len(c["per_rank"]) == 2 and all(p[2] > 0 for p in c["per_rank"])
    spans = [np.load(tmp_path / f"first_{r}.npy") for r in (0, 1)]
    assert spans[0][0] == 0 and spans[0][1] == 4096 and spans[1][0] == 4096 and spans[1][1] == 4096
    exp, _ = oracle_batch([3000 + k for k in range(U)], T, fs, [(1, 3)] * U)
    for r in (0, 1):
        out = np.load(tmp_path / f"out_{r}.npy")
        first, count = int(spans[r][0]), int(spans[r][1])
        want = exp[(np.arange(first, first + count)) % U]
        assert np.array_equal(out, want), f"rank {r}"


def test_bench_self_launch_two_ranks_share_one_gpu():
    """`python bench.py --gpus 2 --share-devices`: started by hand it must re-execute under torch.distributed.run,
    both ranks must run the HIP engine and the JSON must show ranks_seen == 2 with per-rank rates."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--share-devices", "--streams", "4096", "--blocks", "64",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["ranks"]["ranks_seen"] == 2 and d["ranks"]["collective_backend"] == "gloo"
    assert len(d["ranks"]["per_rank_frames_per_s"]) == 2 and all(v > 1e6 for v in d["ranks"]["per_rank_frames_per_s"])
    assert d["config"]["streams_per_gpu"] == 4096 and "configs[1]" in d["config"]["workload"]
    assert d["value"] > 0 and d["roofline"]["kernel_avg_ms"] > 0
    assert d["ranks"]["per_rank_streams"] == [4096, 4096] and len(d["ranks"]["per_rank_device"]) == 2
    assert d["parity"]["ok"] is True and d["parity"]["ranks_ok"] == 2 and d["parity"]["streams"][-1] == 4095


def _bench_json(argv, env=None, timeout=900):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], env=env or _clean_env(), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_counter_collectives_over_rccl():
    """The RCCL code path of `bench.py --gpus N` on the one GPU present: AECM_FORCE_DIST=1 forms a one-rank process group
    with backend nccl (= RCCL) bound to the device, the barrier runs on the device, the counters, the device names and
    the parity flag travel as device tensors through all-reduce / all-gather -- exactly the calls the 8-GPU run makes."""
    env = dict(_clean_env(), AECM_FORCE_DIST="1", MASTER_PORT="29641")
    d = _bench_json(["--gpus", "1", "--dist-backend", "nccl", "--streams", "4096", "--blocks", "64", "--steps", "3", "--warmup", "1",
                     "--no-cpu-baseline"], env=env)
    assert d["ranks"]["collective_backend"] == "nccl" and d["ranks"]["ranks_seen"] == 1 and d["n_gpus"] == 1
    assert len(d["ranks"]["per_rank_device"]) == 1 and "gfx950" in d["ranks"]["per_rank_device"][0]
    assert d["parity"]["ok"] is True and d["parity"]["ranks_ok"] == 1 and d["parity"]["blocks"] == 4 * 64
    assert d["value"] > 1e6


def test_bench_strong_scaling_shards_sum_to_the_total():
    """--total-streams splits a fixed total over the ranks (strong scaling): two ranks sharing the GPU own 4096 + 4095 of
    8191 streams, the frames counted are total x blocks x steps, and every rank's timed workload passes its own parity check."""
    d = _bench_json(["--gpus", "2", "--share-devices", "--total-streams", "8191", "--blocks", "64", "--steps", "3", "--warmup", "1",
                     "--no-cpu-baseline"])
    assert d["scaling"] == "strong" and d["n_gpus"] == 2 and d["ranks"]["ranks_seen"] == 2
    assert d["ranks"]["per_rank_streams"] == [4096, 4095] and sum(d["ranks"]["per_rank_streams"]) == 8191
    assert abs(d["value"] * d["config"]["timed_region_s"] - 8191 * 64 * 3) < 1.0
    assert d["parity"]["ok"] is True and d["parity"]["ranks_ok"] == 2


def test_bench_eight_ranks_preflight_on_one_gpu():
    """The shape of the driver's 8-GPU run (configs[4]) with the ranks mapped onto the one device present: eight processes
    under torch.distributed.run, eight HIP engines, the counter collectives over eight ranks, every rank's own parity check
    (on its share of the host's cores), weak and strong sharding.  tools/scale_preflight.sh runs the same for N = 2, 4, 8."""
    d = _bench_json(["--gpus", "8", "--share-devices", "--streams", "1024", "--blocks", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                    timeout=1500)
    assert d["n_gpus"] == 8 and d["ranks"]["ranks_seen"] == 8 and d["ranks"]["world_size"] == 8
    assert d["ranks"]["per_rank_streams"] == [1024] * 8 and len(set(d["ranks"]["per_rank_device"])) >= 1
    assert d["parity"]["ok"] is True and d["parity"]["ranks_ok"] == 8 and d["parity"]["checker_threads"] >= 1
    assert abs(d["value"] * d["config"]["timed_region_s"] - 8 * 1024 * 64 * 2) < 1.0 and d["scaling"] == "weak"
    d = _bench_json(["--gpus", "8", "--share-devices", "--total-streams", "8191", "--blocks", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                    timeout=1500)
    assert d["scaling"] == "strong" and d["ranks"]["per_rank_streams"] == [1024] * 7 + [1023] and d["parity"]["ranks_ok"] == 8


def test_bench_refuses_more_ranks_than_devices_quickly():
    """`bench.py --gpus 4` on a box with fewer devices must die with ONE clear line before any rendezvous."""
    import time
    import torch
    n = torch.cuda.device_count()
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n + 3)], env=_clean_env(), capture_output=True, text=True, timeout=300)
    dt = time.perf_counter() - t0
    out = r.stdout + r.stderr
    assert r.returncode != 0 and out.count(f"only {n} HIP device(s) visible") == 1, out[-2000:]
    assert dt < 60, dt                       # interpreter + `import torch` only (the 5 s goal is the check itself, not the import)
    # the same launch shape as the driver's (ranks started by torch.distributed.run): every rank refuses before init_process_group
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n + 1), "--master-addr", "127.0.0.1",
                        "--master-port", "29643", str(ROOT / "bench.py"), "--gpus", str(n + 1)], env=_clean_env(), capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode != 0 and f"LOCAL_RANK {n} but only {n} HIP device(s) visible" in out, out[-2000:]


def test_reference_main_cc_unmodified_on_the_gpu(tmp_path):
    """The reference's own caller (main.cc, compiled unmodified against include/ and linked to libaecm_mi355x.so by
    oracle/Makefile: refmain) run on a WAV pair: <near>_out.wav must equal the reference-generated fixture byte for byte."""
    import wave
    from oracle import pyoracle
    from webrtc_aecm_amd.synth import synth_pair
    if not pyoracle.REFMAIN.exists():
        pytest.skip("prebuilt oracle/_ref/aecm_run_refmain not present")
    g = np.load(ROOT / "tests" / "golden" / "session_s7_fs16000_f160_c1_e1_ms40.npz")       # main.cc's parameters
    far, near = synth_pair(int(g["seed"]), int(g["n_blocks"]), 16000, "mixed")
    for name, x in (("far.wav", far), ("near.wav", near)):
        with wave.open(str(tmp_path / name), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes(np.asarray(x, dtype="<i2").tobytes())
    r = subprocess.run([str(pyoracle.REFMAIN), str(tmp_path / "far.wav"), str(tmp_path / "near.wav")], capture_output=True, text=True,
                       input="\n", timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    with wave.open(str(tmp_path / "near_out.wav"), "rb") as w:
        assert w.getframerate() == 16000 and w.getnchannels() == 1
        out = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    n = g["out"].size
    assert np.array_equal(out[:n], g["out"])


def test_cli_equals_the_reference_caller_on_other_sample_formats(tmp_path):
    """WAV pairs in 32-bit float, 24-bit PCM and mu-law: the reference's unmodified main.cc (reading through dr_wav, linked
    to libaecm_mi355x.so) and aecm_run (its own reader) must write the same <near>_out.wav, byte for byte."""
    import shutil
    from helpers import write_wav_format
    from oracle import pyoracle
    from webrtc_aecm_amd import build
    from webrtc_aecm_amd.synth import synth_pair
    if not pyoracle.REFMAIN.exists():
        pytest.skip("prebuilt oracle/_ref/aecm_run_refmain not present")
    build.build()
    far, near = synth_pair(77, 600, 16000, "mixed")
    for fmt_far, fmt_near in (("f32", "f32"), ("s24", "mulaw"), ("f64_", "u8")):
        fmt_far = fmt_far.rstrip("_")
        outs = []
        for who, exe in (("ref", pyoracle.REFMAIN), ("ours", build.CLI)):
            d = tmp_path / f"{who}_{fmt_far}_{fmt_near}"
            d.mkdir()
            write_wav_format(d / "far.wav", 16000, far, fmt_far)
            write_wav_format(d / "near.wav", 16000, near, fmt_near)
            r = subprocess.run([str(exe), str(d / "far.wav"), str(d / "near.wav")], capture_output=True, text=True, input="\n", timeout=300)
            assert r.returncode == 0, (who, fmt_far, fmt_near, r.stdout + r.stderr)
            outs.append((d / "near_out.wav").read_bytes())
        assert len(outs[0]) > 44 + 2 * 160 * 100 and outs[0] == outs[1], (fmt_far, fmt_near)
