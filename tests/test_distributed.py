"""World-size-2 gloo test of the multi-GPU plumbing (sharding + counter gather).  The data path has
no collective: streams are independent, ranks own disjoint static ranges."""
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def test_shard_range_partitions_exactly():
    from webrtc_aecm_amd.dist import shard_range
    for total in (1, 7, 8, 65536, 524288, 1000003):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert spans[-1][0] + spans[-1][1] == total
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_two_rank_gloo_sharded_run_matches_single_process(tmp_path):
    """Each rank processes its shard of 6 streams on the CPU lane simulator (the same DSP source the
    kernel runs) and the counters are gathered; the union of the shards must equal a 1-process run."""
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {str(ROOT)!r}); sys.path.insert(0, {str(ROOT / 'tests')!r})
        import numpy as np, torch
        from webrtc_aecm_amd import dist as adist
        from webrtc_aecm_amd.synth import synth_pair
        import simlib
        rank, local_rank, world = adist.init("gloo")
        first, count = adist.shard_range(6, rank, world)
        t0 = time.perf_counter()
        outs = []
        for s in range(first, first + count):
            far, near = synth_pair(900 + s, 300, 16000)
            outs.append(simlib.SimStream(16000, 1, 3).process(far, near))
        adist.barrier()
        c = adist.gather_counters(count * 300, time.perf_counter() - t0, 1.0 + rank, torch.device("cpu"))
        np.save({str(tmp_path)!r} + f"/out_{{rank}}.npy", np.stack(outs))
        if rank == 0:
            assert c["ranks_seen"] == 2 and [p[0] for p in c["per_rank"]] == [900, 900] and c["backend"] == "gloo"
            open({str(tmp_path)!r} + "/counters.txt", "w").write(f"{{c['frames']}} {{c['kernel_ms']}}")
    """))
    import simlib
    simlib.build()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)], env=env, timeout=300)
    frames, kms = (tmp_path / "counters.txt").read_text().split()
    assert int(frames) == 6 * 300 and float(kms) == 2.0
    got = np.concatenate([np.load(tmp_path / "out_0.npy"), np.load(tmp_path / "out_1.npy")])
    from webrtc_aecm_amd.synth import synth_pair
    for s in range(6):
        far, near = synth_pair(900 + s, 300, 16000)
        assert np.array_equal(got[s], simlib.SimStream(16000, 1, 3).process(far, near))


def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus 2` without WORLD_SIZE re-executes itself under torch.distributed.run with two ranks
    (the driver's command shape for N = 1 must also work for N > 1).  Without a GPU both ranks get as far as the
    device check (made before the rendezvous, so a bad launch dies in seconds) and say so."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: covered by the -m gpu test")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--share-devices", "--streams", "8", "--blocks", "4",
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert out.count("bench.py needs a GPU") >= 2, out[-3000:]          # both ranks ran main() under WORLD_SIZE=2
    assert "WORLD_SIZE=1" not in out


def test_bench_refuses_more_ranks_than_devices_with_one_line():
    """`python bench.py --gpus 2` on a node with fewer HIP devices: one clear line, no rendezvous, no ranks started."""
    import time
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two devices present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    out = r.stdout + r.stderr
    assert r.returncode != 0 and out.count("HIP device(s) visible on this node") == 1, out[-2000:]
    assert "torch.distributed" not in out and time.perf_counter() - t0 < 30
