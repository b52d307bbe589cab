"""GPU tests of the measurement entry points bench.py and tools/ rely on: HIP-event kernel timers, the byte ledger behind
tools/kernel_rooflines.py, the NTT microbenchmark."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_byte_ledger_and_timers_small_ring():
    import image_matching_amd as im
    cc = im.Context(im.default_params(log_n=11, vector_dim=64), 0)
    cc.keygen(3)
    n = 2500
    db = np.random.default_rng(0).integers(-99, 100, size=(n, 64)).astype(np.float64)
    im.DiagonalEnroller(cc, n).serializeDB(db, seed=1)
    q = im.DiagonalReceiver(cc, n).encryptQuery(np.ones(64), seed=2)
    sender = im.DiagonalSender(cc, n)
    sender.indexScenario(q)
    cc.kernel_time_reset()
    im.byte_ledger(1)  # (re)start recording; whatever was there is dropped
    sender.indexScenario(q)
    cc.sync()
    led = im.byte_ledger(0)
    tens = {k: v for k, v in led.items() if k.startswith("k_hydia_tensor")}
    assert tens and sum(v[0] for v in tens.values()) in (1, 2)
    # loop B's ledger entry = resident database + rotated queries + accumulators of the launches
    assert sum(v[1] for v in tens.values()) >= cc.db_stats()[2]
    assert led["k_tensor<false>"][0] >= 10 and all(b > 0 for _, b in led.values())
    assert im.byte_ledger(-1) == {}  # stopped and cleared
    ms, launches = cc.kernel_time("hydia_tensor")
    assert launches == 1 and ms > 0
    ms, launches = cc.kernel_time("ks_inner_product")
    assert launches >= 20 and ms > 0
    del q
    cc.close()


def test_ntt_microbenchmark_entry_point():
    import image_matching_amd as im
    cc = im.Context()
    for inv in (False, True):
        assert cc.bench_ntt(4, 1, 3, inv, 2) > 0      # FP64 limbs
        assert cc.bench_ntt(2, 12, 4, inv, 2) > 0     # 60-bit special primes
    with pytest.raises(im.HydiaError):
        cc.bench_ntt(1, 15, 4, False, 1)              # moduli out of range
    cc.close()


def test_op_ledger_prices_a_relinearisation_like_the_survey():
    """bench.py's roofline.step sums the ledger's op:* entries: SURVEY 8d's prices (relinearise 80 + rescale 24 limb-transforms of
    2 N 8 B at 12 limbs + the 24 MiB key; loop B = resident database + rotated queries + accumulators)"""
    import image_matching_amd as im
    cc = im.Context()
    cc.fill_eval_keys_random(1)
    n = 16384
    cc.set_matvec("hoisted")  # the reference's form: 511 hoisted rotations, one relinearisation per block
    cc.db_fill_random(n, 2)
    rng = np.random.default_rng(0)
    q = np.stack([rng.integers(0, int(m), size=(2, cc.N), dtype=np.uint64) for m in cc.moduli[:cc.nQ]], axis=1)
    gq = cc.import_ct(q, cc.delta)
    sender = im.DiagonalSender(cc, n)
    sender.computeSimilarity(gq)
    im.byte_ledger(1)
    sender.computeSimilarity(gq)
    cc.sync()
    led = im.byte_ledger(0)
    ops = {k: v for k, v in led.items() if k.startswith("op:")}
    lp = cc.N * 8
    assert cc.db_kind() == 5
    assert ops["op:relin_rescale"] == (1, (80 + 24) * 2 * lp + 3 * 2 * 16 * lp)
    assert ops["op:loop_b"][1] == cc.db_stats()[2] + (512 * 2 + 3) * 12 * lp
    assert abs(ops["op:loop_a"][1] - 13.9e9) < 0.6e9  # the review's figure for loop A: keys in once, rotations out once
    assert all(v[1] > 0 for v in ops.values())
    del gq
    cc.close()


def test_bench_gpus_flag_runs_two_real_ranks_on_this_gpu():
    """`python bench.py --gpus 2` with no launcher: bench.py starts the two ranks itself.  Rehearsal mode (both ranks compute on GPU 0,
    collectives over gloo) is the only way to run world 2 on a one-GPU box; the line must say how many ranks really joined."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HYDIA_BENCH_REHEARSE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--total-log2n", "15", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-secondary-weak"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_initialised"] == 2 and out["collective_backend"] == "gloo"
    assert out["config"]["result_correct"] is True and out["scaling"] == "strong"
    rf = out["roofline"]
    assert 0 < rf["frac"] <= 1.0 and rf["algorithmic_frac"] >= rf["frac"]
    assert rf["step"]["inherent_bytes"] > rf["bytes_per_launch"] and 0 < rf["step"]["frac"] < rf["frac"] + 1e-9
    # round 5: an N > 1 line explains itself — communication / compute per rank (max and min), the loop-A mode, every phase of the
    # sharded sender, and the phases of rank 0 add up to the instrumented step
    sp = out["step_split"]
    assert set(out["comm_ms"]) == {"max", "min"} and set(out["compute_ms"]) == {"max", "min"} and out["compute_ms"]["max"] > 0
    assert out["loop_a_mode"] in ("local", "split", "replicated")
    assert {"query_broadcast", "form_check", "local_matvec_comparator", "result_gather"} <= set(sp["phases_ms"]) and len(sp["per_rank_ms"]) == 2
    assert abs(sp["rank0_sum_ms"] - sp["instrumented_ms_per_step"]) <= 0.10 * sp["instrumented_ms_per_step"], sp
    assert "model" in out  # (2^15 in all is not a modelled configuration: null; tests/test_bench_launcher.py checks the model itself)
    assert rf["stream_ceiling"]["source"].startswith("profiles/") and rf["vs_measured_stream_ceiling"] > 0
