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
