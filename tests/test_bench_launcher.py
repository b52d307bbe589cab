"""`python bench.py --gpus N` without a launcher around it has to start N ranks ITSELF (the driver's command shape; round 3's
bench ignored the flag and measured one GPU).  CPU tests of that plumbing through the real entry point: the children are
subprocesses started before any GPU call, rendezvous over gloo on 127.0.0.1, rank 0's JSON line is relayed, a failing rank's
exit code comes back and the other ranks are stopped, and a launcher whose world size is not the --gpus asked for is refused.
The engine itself needs a GPU: tests/test_gpu_measure.py runs the same entry with two real ranks on the one-GPU box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_flag_starts_that_many_ranks():
    for n in (2, 3):
        r = run(["--gpus", str(n), "--selftest-launcher"])
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout  # stdout carries ONE line
        out = json.loads(lines[0])
        assert out["n_gpus"] == n and out["ranks_initialised"] == n and out["self_launched"] is True
        assert out["rank_sum"] == n * (n + 1) / 2  # every rank took part in the collective


def test_failing_rank_stops_the_job_with_its_exit_code():
    r = run(["--gpus", "2", "--selftest-launcher"], HYDIA_BENCH_SELFTEST_FAIL_RANK="1")
    assert r.returncode == 7
    assert "rank 1 exited with code 7" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]  # no result line from a broken job


def test_world_size_mismatch_is_refused():
    r = run(["--gpus", "2", "--selftest-launcher"], WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    assert r.returncode == 2 and "refusing" in r.stderr
    r = run(["--gpus", "1", "--selftest-launcher"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    assert r.returncode == 2 and "refusing" in r.stderr


def test_external_launcher_still_works():
    """the driver's N > 1 command shape: torch.distributed.run starts the ranks, bench.py must not start more"""
    import socket
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    with socket.socket() as sk:  # a free port, not a fixed one (back-to-back runs on one box collided on a port in TIME_WAIT)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--selftest-launcher"], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["ranks_initialised"] == 2 and out["self_launched"] is False


def test_more_ranks_than_gpus_is_refused_before_anything_starts():
    """no GPU in this container: a real (non-rehearsal, non-selftest) multi-GPU run must say so and exit 2 without starting ranks"""
    import pytest
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this machine has the GPUs: nothing to refuse")
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr


def test_step_model_and_stream_ceiling_come_from_committed_profiles():
    """bench.py --gpus N prints DESIGN section 7's prediction beside the measured step: from profiles/scaling_model.json, for the two
    modelled databases (BASELINE configs 5 and 4); the stream ceiling is parsed from the newest committed stream_rate.txt"""
    sys.path.insert(0, ROOT)
    import bench
    for lg, world in ((20, 8), (20, 2), (17, 4)):
        for mode in ("local", "replicated", "split"):
            m = bench.model_step(lg, world, mode)
            assert m["mode"] == mode and m["unmeasured_on_multi_gpu_hardware"] is True
            assert abs(m["predicted_ms_per_step"] - (m["terms"]["compute_ms"] + m["terms"]["comm_ms"])) < 1e-3
    assert bench.model_step(20, 8, "local")["predicted_ms_per_step"] < bench.model_step(20, 2, "local")["predicted_ms_per_step"]
    assert bench.model_step(15, 2, "local") is None  # not a modelled configuration
    gbs, src = bench.stream_ceiling()
    assert 6000 < gbs < 8000 and src.startswith("profiles/r") and src.endswith("stream_rate.txt")


def test_straggler_after_rank0_is_stopped(tmp_path):
    """rank 0 finishes, another rank hangs: the launcher stops it after the grace period and the run fails"""
    r = run(["--gpus", "2", "--selftest-launcher"], HYDIA_BENCH_SELFTEST_HANG_RANK="1", HYDIA_BENCH_STRAGGLER_S="3")
    assert r.returncode == 5 and "still run" in r.stderr
