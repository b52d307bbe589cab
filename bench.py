#!/usr/bin/env python3
"""bench.py — encrypted DB vectors matched/sec (HyDia, CKKS N = 2^15) on N MI355X of one node.

A "step" is ONE query through DiagonalSender::indexScenario (/root/reference/src/sender/sender_diag.cpp:52-63: 511 hoisted
rotations, per block 512 tensor products + 1 relinearise + 1 rescale, degree-59 Chebyshev o f4 comparator) over the
encrypted database resident in HBM.
  N = 1   the 2^20-vector database (64 blocks, 142.5 GiB resident) on one GPU — the configuration BASELINE.json quotes its target on.
  N > 1   STRONG scaling by default: the SAME 2^20-vector database sharded by 16384-vector row-blocks over the N ranks
          (BASELINE config 5 at N = 8: 8 blocks per GPU; `--total-log2n 17` at N = 4 is config 4), through
          image_matching_amd.sharding.DistDiagonalSender — the class the tests pin bit-exactly against one context: rank 0's query
          is broadcast, loop A's rotations are either shared out over the ranks and all-gathered over xGMI (SURVEY 8e option B) or
          recomputed by every rank (option A) — whichever the warm-up measures faster on this node —, every rank runs an independent
          mat-vec + comparator on its own blocks, and the result ciphertexts are gathered over RCCL to rank 0 in global block order.
          `--weak` keeps 2^log2n vectors PER GPU instead; its figure also rides in config.secondary of a default run.
Data is synthetic with the distribution of the reference's tools/gen_dataset.sh (query = ones, random rows in [-99,99], planted
matches in {1,2,3}); the database is REAL ciphertexts produced by the on-GPU enroller, and after the timed region rank 0
decrypts the gathered index result and checks it against the planted matches (global indices).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

# The engine runs two comparator lanes (HIP streams); torch and RCCL bring their own.  With ROCm's default of 4 hardware queues
# the lanes end up sharing a queue with RCCL's streams and serialise (+4 ms per query, measured with tools/probe_dist.py), so ask
# for 8 before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20250725
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PUBLISHED_2P20_INDEX_VPS = (1 << 20) / 96.52172  # /root/reference/tools/figures/approach5.csv:12 (Xeon Gold 5412U, 48 threads)


def synth_rows(lo, hi, dim, planted):
    """Rows [lo, hi) of the synthetic database (tools/gen_dataset.sh distribution: uniform integers in [-99, 99]; matching rows
    uniform in {1,2,3}).  Generated per 16384-row block from a seed that depends on the GLOBAL block index, so any sharding of
    the same database enrols the same vectors."""
    out = np.empty((hi - lo, dim), dtype=np.float64)
    S = 16384
    for b in range(lo // S, -(-hi // S)):
        rng = np.random.default_rng(SEED + 1000003 * b)
        blk = rng.integers(-99, 100, size=(S, dim), dtype=np.int8).astype(np.float64)
        for i in planted:
            if b * S <= i < (b + 1) * S:
                blk[i - b * S] = rng.integers(1, 4, size=dim)
        a, z = max(lo, b * S), min(hi, (b + 1) * S)
        out[a - lo:z - lo] = blk[a - b * S:z - b * S]
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_share():
    """CPUs this job may really use: the scheduler affinity, capped by the cgroup CPU quota when there is one (a GPU box shows
    every hardware thread of its host but grants a share of them)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    cores = aff if quota is None else max(1, min(aff, int(math.ceil(quota))))
    return cores, aff, quota


def cpu_baseline():
    """Time the CPU oracle (this repository's restatement of the reference algorithm — OpenFHE itself is absent, kind "port")
    on this host's cores, the way BASELINE.md §2 prescribes: the same indexScenario at G = 1 and G = 3 blocks in RAM, the
    marginal seconds per extra 16384-vector block, the linear extrapolation to the 64 blocks of 2^20 vectors (the published
    curve is linear in blocks, tools/figures/approach5.csv), plus one G = 1 query whose loop B re-reads every ciphertext from
    disk like the reference's timed loop (sender_diag.cpp:87-91).  Only this leg touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    # all host cores this job has: the affinity mask, capped by the cgroup quota — NOT the test helper's cap of 16
    cores, aff, quota = cpu_share()
    O.lib().hyo_set_num_threads(cores)
    P = O.Params()
    K = O.Keys(P, SEED)
    Or = O.Oracle(P, K)
    S, n3 = P.slots, 3 * P.slots
    planted = [5, S + 9, n3 - 1]
    db = synth_rows(0, n3, P.dim, planted)
    dbc = Or.enroll(db, 99, matvec="hoisted")  # the reference's own algorithm (sender_diag.cpp:22-26), whatever form the GPU run chose
    q = Or.encrypt_query(np.ones(P.dim), 5, 1)

    def timed(fn):
        t0 = time.time()
        r = fn()
        return time.time() - t0, r
    t_rot, rot = timed(lambda: Or.rotate_query(q))
    del rot
    t1, idx1 = timed(lambda: Or.index_scenario(q, dbc, S))
    ok = Or.decrypt_index(idx1) == [5]
    t3, idx3 = timed(lambda: Or.index_scenario(q, dbc, n3))
    ok = ok and Or.decrypt_index(idx3) == planted
    marginal = (t3 - t1) / 2.0
    t_2p20 = t1 + 63 * marginal
    disk = None
    d = tempfile.mkdtemp(prefix="hydia_serial_")
    try:
        free = shutil.disk_usage(d).free
        if free > 5 << 30:
            sub = O.CtArray(P, dbc.h, P.dim)  # the first block's 512 ciphertexts (a view: never freed through `sub`)
            try:
                Or.write_db_files(sub, d)
            finally:
                sub.h = None
            td, idxd = timed(lambda: Or.index_scenario_files(q, d, S))
            ok = ok and Or.decrypt_index(idxd) == [5]
            disk = {"s_per_query_G1": round(td, 3), "vectors_per_s_G1": round(S / td, 1),
                    "note": "loop B reads index<t>.bin (6 MiB each, page cache warm) inside the parallel loop"}
    except Exception as e:  # no scratch space on this box: the RAM figure stands alone
        disk = {"error": str(e)[:200]}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    cores = int(O.lib().hyo_num_threads())
    return {
        "value": (1 << 20) / t_2p20, "unit": "vectors/s", "cores": cores, "kind": "port",
        "cpu_model": cpu_model(), "omp_max_threads": cores, "host_logical_cpus": os.cpu_count(), "affinity_cpus": aff,
        "cgroup_cpu_quota": quota,
        "sample": "oracle indexScenario at N=2^15 timed at G=1 (%.2f s, of which the 511 hoisted rotations %.2f s) and G=3 (%.2f s) "
                  "blocks of 16384 vectors, DB in RAM: marginal %.3f s per block; value = 2^20 / (t_G1 + 63 x marginal) = linear "
                  "extrapolation to the 64 blocks of the GPU workload (%.1f s per query); results %s"
                  % (t1, t_rot, t3, marginal, t_2p20, "correct" if ok else "WRONG"),
        "s_per_query": {"G1": round(t1, 3), "G3": round(t3, 3), "marginal_per_block": round(marginal, 3),
                        "extrapolated_2p20": round(t_2p20, 2)},
        "vectors_per_s_measured": {"G1": round(S / t1, 1), "G3": round(n3 / t3, 1)},
        "disk_reread_variant": disk,
        "published_anchor": {"vectors_per_s": round(PUBLISHED_2P20_INDEX_VPS, 1),
                             "what": "reference's own 2^20 index computation, 96.52 s on Xeon Gold 5412U / 48 threads incl. disk reads "
                                     "(tools/figures/approach5.csv:12)"},
    }


def git_head():
    """commit the numbers belong to: git when there is a checkout, else the VERSION file __graft_entry__.build() leaves in the tree
    (the GPU box receives a snapshot without .git)"""
    try:
        h = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
        if h:
            return h
    except Exception:
        pass
    try:
        return open(os.path.join(ROOT, "image_matching_amd", "VERSION")).read().strip()
    except OSError:
        return ""


def stream_ceiling(bits46=True):
    """what this GPU gives a read-once sequential stream in loop B's own access pattern (tools/ubench/stream_rate.hip), from the newest
    committed profiles/r*/stream_rate.txt: the 46-bit-unit figure for a 46-bit database ("... with 46-bit residues ...: X TB/s of distinct
    bytes"), else the 48-bit "workgroup-sequential layout" one -> (GB/s, file), or (None, None)"""
    import glob
    import re
    pats = ([r"with 46-bit residues[^\n]*?:\s*([0-9.]+) TB/s"] if bits46 else []) + [r"^\s*workgroup-sequential layout:\s*([0-9.]+) TB/s"]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "stream_rate.txt")), reverse=True):
        try:
            text = open(path).read()
        except OSError:
            continue
        for pat in pats:
            m = re.search(pat, text, re.M)
            if m:
                return float(m.group(1)) * 1e3, os.path.relpath(path, ROOT)
    return None, None


def model_step(log2n_total, world, mode):
    """DESIGN.md section 7's prediction of one multi-GPU step from the committed ONE-GPU components (profiles/scaling_model.json):
    what the first SCALE run is to be compared with, term by term.  mode: "local" (the auto rule's baby-step / giant-step split, a short
    loop A per rank), "replicated" (every rank recomputes the 511 rotations), "split" (loop A shared out + all-gather of 3 GiB)."""
    try:
        m = json.load(open(os.path.join(ROOT, "profiles", "scaling_model.json")))
        row = m["log2n"][str(log2n_total)][str(world)]
    except (OSError, KeyError, ValueError):
        return None
    g = m["gather_sync_ms"]
    out = {"source": "profiles/scaling_model.json (" + m["source"] + ")", "mode": mode, "unmeasured_on_multi_gpu_hardware": True}
    if mode == "local" and "whole_query_auto_split_ms" in row:
        terms = {"compute_ms": row["whole_query_auto_split_ms"], "comm_ms": g}
    elif mode == "split":
        ag = {k: (world - 1) / world * m["rotations_bytes"] / max(world - 1, 1) / (v * 1e9) * 1e3 for k, v in m["xgmi_link_GBs"].items()}
        terms = {"compute_ms": row["loop_a_share_ms"] + row["rest_on_given_rotations_ms"], "comm_ms": g + ag["high"],
                 "loop_a_share_ms": row["loop_a_share_ms"], "rest_on_given_rotations_ms": row["rest_on_given_rotations_ms"],
                 "rotations_all_gather_ms_at_153_GBs_per_link": round(ag["high"], 3), "rotations_all_gather_ms_at_76.8_GBs_per_link": round(ag["low"], 3)}
    else:
        terms = {"compute_ms": row["whole_query_replicated_loop_a_ms"], "comm_ms": g}
    terms["gather_and_sync_ms"] = g
    out["terms"] = terms
    out["predicted_ms_per_step"] = round(terms["compute_ms"] + terms["comm_ms"], 3)
    return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves.  The children are plain
    subprocesses of this interpreter (never os.exec*); this process only ever COUNTS devices (torch.cuda.device_count(), which creates
    no context) and never execs, one child per LOCAL_RANK, rendezvous on 127.0.0.1.  Rank 0's stdout is captured and its JSON line
    relayed; the other ranks' stdout goes to stderr.  The first rank that fails takes the others down (by exact PID) and its exit code
    is returned; once rank 0 has finished the others get HYDIA_BENCH_STRAGGLER_S seconds (default 120) to follow, then are stopped
    and the run counts as failed (code 5)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HYDIA_BENCH_SELF_LAUNCHED="1")
        # like torch.distributed.run: one OpenMP thread per rank unless the caller says otherwise — N ranks that each spin up a thread
        # per host core oversubscribe the CPU (the gloo rehearsal's host staging ran 8x slower: 368 vs 47 ms per step)
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, cwd=os.getcwd()))
    rc, out0 = 0, b""
    try:
        import threading
        buf = []
        rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
        rd.start()
        live = set(range(n))
        grace, rank0_done = float(os.environ.get("HYDIA_BENCH_STRAGGLER_S", "120")), None
        while live and rc == 0:
            for r in sorted(live):
                c = procs[r].poll()
                if c is not None:
                    live.discard(r)
                    if r == 0:
                        rank0_done = time.time()
                    if c != 0:
                        rc = c if c > 0 else 128 - c
                        sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, c))
                        break
            if rc == 0 and live and rank0_done is not None and time.time() - rank0_done > grace:
                rc = 5
                sys.stderr.write("bench.py: rank 0 finished %.0f s ago and rank(s) %s still run; stopping them\n" % (grace, sorted(live)))
            time.sleep(0.05)
        if rc:
            for r in live:
                procs[r].terminate()
            for r in live:
                try:
                    procs[r].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
        rd.join(timeout=30)
        out0 = buf[0] if buf else b""
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    lines = [ln for ln in out0.decode(errors="replace").splitlines() if ln.startswith("{")]
    if rc == 0 and not lines:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        rc = 4
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return rc


def launcher_selftest(world, rank):
    """--selftest-launcher: what a rank does when only the launch plumbing is to be exercised (CPU test of `--gpus N`): join the gloo
    group the launcher described, all-reduce the ranks, rank 0 prints who came."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    fail = os.environ.get("HYDIA_BENCH_SELFTEST_FAIL_RANK")
    if fail is not None and int(fail) == rank:
        sys.exit(7)  # before the barrier: the other ranks hang in it until the launcher stops them
    dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": "launcher", "n_gpus": world, "ranks_initialised": dist.get_world_size(),
                          "backend": dist.get_backend(), "rank_sum": float(t.item()),
                          "self_launched": os.environ.get("HYDIA_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
    if os.environ.get("HYDIA_BENCH_SELFTEST_HANG_RANK") == str(rank) and rank != 0:
        time.sleep(3600)  # (test of the launcher's straggler deadline)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=20,
                    help="log2 of the DB vectors (one GPU: of that GPU; N > 1: of the TOTAL database unless --weak)")
    ap.add_argument("--total-log2n", type=int, default=None,
                    help="N > 1: log2 of the TOTAL DB vectors, split over the ranks by blocks (default 20 = BASELINE config 5; 17 = config 4)")
    ap.add_argument("--weak", action="store_true", help="N > 1: 2^log2n vectors PER GPU (weak scaling) instead of a fixed total")
    ap.add_argument("--loop-a", choices=["auto", "split", "replicated"], default="auto",
                    help="N > 1: loop A's 511 rotations shared out over the ranks and all-gathered (split), recomputed by every rank "
                         "(replicated), or whichever the warm-up measures faster (auto)")
    ap.add_argument("--no-secondary-weak", action="store_true", help="N > 1: skip the weak-scaling figure of config.secondary")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--random-db", action="store_true", help="fill the DB with random residues instead of enrolling")
    ap.add_argument("--selftest-launcher", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    # --gpus N is the contract: without a launcher around us (no WORLD_SIZE) we start the N ranks ourselves; under one
    # (torch.distributed.run) its world size has to be the N that was asked for — a mismatch would silently measure another job
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if not args.selftest_launcher and os.environ.get("HYDIA_BENCH_REHEARSE") != "1":
            import torch  # (counting devices does not initialise the GPU)
            if torch.cuda.device_count() < args.gpus:
                sys.stderr.write("bench.py: --gpus %d but %d GPU(s) visible on this node (RCCL wants one GPU per rank; HYDIA_BENCH_REHEARSE=1 "
                                 "runs every rank on GPU 0 over gloo, for control-flow rehearsal only)\n" % (args.gpus, torch.cuda.device_count()))
                sys.exit(2)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks; refusing to run a different job than asked\n"
                         % (args.gpus, os.environ.get("WORLD_SIZE")))
        sys.exit(2)
    if args.selftest_launcher:
        return launcher_selftest(int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")))

    # stdout carries ONE line, the JSON result.  Native libraries write there too (RCCL prints a five-line version banner on fd 1 when
    # its first communicator comes up), so fd 1 is pointed at stderr for the whole run and the result goes to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = torch = None
    # HYDIA_BENCH_REHEARSE=1: every rank computes on GPU 0 and the collectives go through gloo/host memory — lets the multi-rank
    # control flow be exercised on a one-GPU box (never used for reported numbers)
    rehearse = os.environ.get("HYDIA_BENCH_REHEARSE") == "1"
    # HYDIA_BENCH_FORCE_DIST=1 (under torchrun --nproc-per-node 1): take the multi-rank path with a one-rank RCCL group — exercises
    # the real nccl init / collectives / device-pointer plumbing on a one-GPU box
    multi = world > 1 or os.environ.get("HYDIA_BENCH_FORCE_DIST") == "1"
    if multi:
        import torch  # noqa: F811  (device memory + RCCL only)
        import torch.distributed as dist  # noqa: F811
        if rehearse:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import image_matching_amd as im
    cc = im.Context(im.default_params(), 0 if rehearse else local_rank)
    strong = world > 1 and not args.weak
    if world == 1:
        n_total = 1 << (args.total_log2n if args.total_log2n is not None else args.log2n)
    elif strong:
        n_total = 1 << (args.total_log2n if args.total_log2n is not None else args.log2n)
    else:
        n_total = world * (1 << args.log2n)
    dim, S = cc.dim, cc.slots
    first, last = im.shard_vectors(n_total, S, world, rank)
    n_local = last - first
    G_local = -(-n_local // S)

    t0 = time.time()
    cc.keygen(SEED)  # same seed on every rank -> identical keys, no key distribution needed
    t_keygen = time.time() - t0
    planted = sorted(set([0, n_total // 2, n_total - 1])) if n_total > 2 else [0]
    t0 = time.time()
    if args.random_db:
        if n_local:
            cc.set_matvec(im.group_babies(cc, n_total, world) if world > 1 else "auto")  # ONE split for every shard: the group-wide rule
            cc.db_fill_random(n_local, SEED + rank)
            cc.set_matvec("auto")
    else:
        rows = synth_rows(first, last, dim, planted)
        im.DistDiagonalEnroller(cc, n_total, rank, world).serializeDB(rows, seed=SEED + 7)
        del rows
    t_enroll = time.time() - t0
    receiver = im.DiagonalReceiver(cc, n_total)
    qc = receiver.encryptQuery(np.ones(dim), seed=SEED, nonce=1) if rank == 0 else None
    if multi:
        sender = im.DistDiagonalSender(cc, n_total, dist, rank, world, staging="host" if rehearse else "device")
    else:
        sender = im.DiagonalSender(cc, n_total)

    def fence():
        cc.sync()
        if multi:
            if not rehearse:
                torch.cuda.synchronize()
            dist.barrier()
            if not rehearse:
                torch.cuda.synchronize()

    if multi and not rehearse:
        # open the RCCL channels outside the timed region even when --warmup 0 (lazy connection set-up takes seconds; it is
        # communicator start-up, not part of a query)
        probe = torch.zeros(1024, dtype=torch.int64, device="cuda")
        dist.broadcast(probe, src=0)
        dist.gather(probe, [torch.empty_like(probe) for _ in range(world)] if rank == 0 else None, dst=0)
        torch.cuda.synchronize()
    res = None

    def timed(snd, query, k):
        """k steps bracketed by fences; the MAX over ranks of the wall time"""
        nonlocal res
        fence()
        t_0 = time.time()
        for _ in range(k):
            res = snd.indexScenario(query)
        fence()
        dt = time.time() - t_0
        if multi:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    # loop A across ranks (untimed set-up): both forms give the same ciphertexts; which is faster depends on what the node's xGMI
    # all-gather of the 3 GiB of rotated queries costs against recomputing them — measured here, on this node, then fixed
    loop_a = None
    if multi and world > 1 and getattr(sender, "bsgs", False):
        # few blocks per GPU: the database is enrolled for a short loop A (B - 1 hoisted rotations, giant steps per block): nothing to share out
        loop_a = {"mode": "local", "why": "baby-step / giant-step split with %d babies: every rank computes its own %d rotations" % (cc.db_babies() or cc.auto_babies(max(G_local, 1)), (cc.db_babies() or cc.auto_babies(max(G_local, 1))) - 1)}
        sender.rotation_split = False
    elif multi and world > 1:
        loop_a = {"mode": args.loop_a}
        if args.loop_a == "auto":
            trial = {}
            for mode in ("split", "replicated"):
                sender.rotation_split = mode == "split"
                timed(sender, qc, 1)  # buffers, channels
                trial[mode] = timed(sender, qc, 2) / 2 * 1e3
            loop_a.update({"ms_per_step_split": round(trial["split"], 3), "ms_per_step_replicated": round(trial["replicated"], 3)})
            loop_a["mode"] = "split" if trial["split"] <= trial["replicated"] else "replicated"
        sender.rotation_split = loop_a["mode"] == "split"
    for _ in range(args.warmup):
        res = sender.indexScenario(qc)
    fence()
    cc.kernel_time_reset()
    elapsed = timed(sender, qc, args.steps)

    ms_tensor, launches = cc.kernel_time("hydia_tensor")
    # N > 1: where a step goes, per rank (outside the timed region: the phases are fenced, which the timed steps are not)
    split_report = None
    if multi:
        k_i = 3
        sender.timing = {}
        dt_i = timed(sender, qc, k_i)
        mine = {k: v / k_i for k, v in sender.timing.items() if k != "calls"}
        sender.timing = None
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        if rank == 0:
            comm = [sum(d.get(k, 0.0) for k in sender.COMM_PHASES) for d in per_rank]
            comp = [sum(d.get(k, 0.0) for k in sender.COMPUTE_PHASES) for d in per_rank]
            names = sorted({k for d in per_rank for k in d})
            split_report = {
                "what": "%d further steps with every phase of DistDiagonalSender fenced (library stream + torch stream) and timed on the host, "
                        "per rank; comm = query_broadcast + form_check + rotations_all_gather + result_gather (a rank's wait for slower ranks "
                        "is inside its collectives), compute = loop_a_share + local_matvec_comparator" % k_i,
                "instrumented_ms_per_step": round(dt_i / k_i * 1e3, 3),
                "comm_ms": {"max": round(max(comm), 3), "min": round(min(comm), 3)},
                "compute_ms": {"max": round(max(comp), 3), "min": round(min(comp), 3)},
                "phases_ms": {k: {"max": round(max(d.get(k, 0.0) for d in per_rank), 3), "min": round(min(d.get(k, 0.0) for d in per_rank), 3)} for k in names},
                "rank0_sum_ms": round(comm[0] + comp[0], 3),
                "per_rank_ms": [{k: round(v, 3) for k, v in d.items()} for d in per_rank]}
    # secondary figure of SURVEY 8d (outside the timed region): computeSimilarity alone = loop A + loop B + relin + rescale
    ms_similarity = None
    if not multi:
        cc.sync()
        t1 = time.time()
        for _ in range(3):
            sim = sender.computeSimilarity(qc)
        cc.sync()
        ms_similarity = (time.time() - t1) / 3 * 1e3
        # membershipScenario (sender_diag.cpp:35-50) = indexScenario + EvalAddMany + EvalSum: the other published column
        t1 = time.time()
        for _ in range(3):
            mem = sender.membershipScenario(qc)
        cc.sync()
        ms_membership = (time.time() - t1) / 3 * 1e3
        membership_ok = None if args.random_db else bool(receiver.decryptMembership(mem))
        del mem
        del sim
    # correctness of what was just timed: rank 0 holds every block's result in global block order -> global indices
    correct = True
    if not args.random_db and rank == 0 and res is not None:
        correct = receiver.decryptIndex(res) == planted and len(res) == -(-n_total // S)

    # everything that describes the database that was TIMED is captured here: the optional weak figure below re-enrols on the same context
    db_resident_bytes = cc.db_stats()[2]
    db_kind_timed, db_babies_timed, db_group_timed, db_bits_timed = cc.db_kind(), cc.db_babies(), cc.db_group(), cc.db_residue_bits()
    ranks_initialised = dist.get_world_size() if multi else 1
    backend = dist.get_backend() if multi else None
    # inherent bytes of the whole step (outside the timed region): one more query with the byte ledger on; the evaluator records what
    # each OPERATION has to move as SURVEY 8d prices it ("op:*" entries: loop B's resident bytes, loop A's keys in + rotations out,
    # 2 N 8 B per limb-transform of every relinearisation / rescale / rotation, each evaluation key once per launch, ct x ct operands)
    step_ops = None
    if rank == 0 or multi:
        im.byte_ledger(1)
        _r = sender.indexScenario(qc)
        cc.sync()
        led = im.byte_ledger(0)
        del _r
        step_ops = {k[3:]: v[1] for k, v in led.items() if k.startswith("op:")}
    if multi:  # every rank takes the same branches below (collectives inside)
        flag = [bool(correct)]
        dist.broadcast_object_list(flag, src=0)
        correct = flag[0]
    # N > 1, default run: the weak-scaling figure (2^log2n vectors PER GPU) beside the strong one, outside the timed region
    weak = None
    if multi and world > 1 and strong and not args.no_secondary_weak and not args.random_db and correct:
        try:
            n_w = world * (1 << args.log2n)
            f_w, l_w = im.shard_vectors(n_w, S, world, rank)
            planted_w = sorted(set([0, n_w // 2, n_w - 1]))
            rows = synth_rows(f_w, l_w, dim, planted_w)
            im.DistDiagonalEnroller(cc, n_w, rank, world).serializeDB(rows, seed=SEED + 7)
            del rows
            snd_w = im.DistDiagonalSender(cc, n_w, dist, rank, world, staging="host" if rehearse else "device",
                                          rotation_split=sender.rotation_split)
            q_w = im.DiagonalReceiver(cc, n_w).encryptQuery(np.ones(dim), seed=SEED, nonce=1) if rank == 0 else None
            keep = res
            timed(snd_w, q_w, 1)
            dt = timed(snd_w, q_w, 3)
            ok_w = im.DiagonalReceiver(cc, n_w).decryptIndex(res) == planted_w if rank == 0 else True
            res = keep
            weak = {"db_vectors_total": n_w, "db_vectors_per_gpu": 1 << args.log2n, "ms_per_step": round(dt / 3 * 1e3, 3),
                    "vectors_per_s": round(n_w * 3 / dt), "result_correct": bool(ok_w)}
            del snd_w, q_w
        except Exception as e:  # the headline line must not die on the optional figure
            weak = {"error": repr(e)[:300]}

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        nl, N = cc.nQ, cc.N
        bsgs = db_kind_timed == 6
        # loop B's operands: the database once, the rotated queries once, the degree-2 accumulators once.  Hoisted form: dim rotated
        # queries, one accumulator per block; baby-step / giant-step form: B = 32 babies, dim / B accumulators per block
        n_rot = db_babies_timed if bsgs else dim
        n_acc = G_local * (dim // n_rot if bsgs else 1)
        algo_bytes = (G_local * dim + n_rot) * 2 * nl * N * 8 + n_acc * 3 * nl * N * 8
        avg_launch_s = ms_tensor / max(launches, 1) / 1e3
        achieved = algo_bytes / avg_launch_s / 1e9 if launches else 0.0
        # bytes the kernel has to move given the RESIDENT layout (48-bit residues for the 45/46-bit limbs of the database;
        # rotated queries and accumulators at 8 bytes): what the wire sees when nothing is read twice
        resident_bytes = db_resident_bytes + n_rot * 2 * nl * N * 8 + n_acc * 3 * nl * N * 8
        wire = resident_bytes / avg_launch_s / 1e9 if launches else 0.0
        traffic = traffic_meta = None
        tpath = os.path.join(ROOT, "profiles", "tensor_traffic.json")
        knobs = [k for k in ("HYDIA_DB_UNPACKED", "HYDIA_DB_CT_MAJOR", "HYDIA_DB_48BIT", "HYDIA_TENSOR_BPP", "HYDIA_TENSOR_NW") if os.environ.get(k)]
        if os.path.exists(tpath) and not args.random_db and not knobs and world == 1:
            try:
                tj = json.load(open(tpath))
                if tj.get("log2n") == n_total.bit_length() - 1:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_meta = {"profiled_at_commit": tj.get("commit"), "source": tj.get("source"),
                                    "kernel_source_sha": tj.get("kernel_sha"), "stale": tj.get("kernel_sha") != kernel_sha()}
                    if traffic_meta["stale"]:
                        traffic = None  # the loop-B kernel changed since the counters were collected: do not quote them
            except Exception:
                traffic = None
        db_gib = db_resident_bytes / 2 ** 30
        # whole-step roofline: where the time outside loop B sits
        step_roofline = None
        if step_ops:
            tot = sum(step_ops.values())
            loop_b_b, loop_a_b = step_ops.get("loop_b", 0.0), step_ops.get("loop_a", 0.0)
            step_roofline = {
                "inherent_bytes": tot, "achieved": tot / (ms_step * 1e-3) / 1e9, "unit": "GB/s", "frac": tot / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "ms_per_step": ms_step, "ms_at_peak": tot / (HBM_PEAK_GBS * 1e9) * 1e3,
                "bytes_by_operation": {k: round(v) for k, v in sorted(step_ops.items(), key=lambda kv: -kv[1])},
                "ms_loop_b": avg_launch_s * 1e3, "ms_outside_loop_b": ms_step - avg_launch_s * 1e3,
                "outside_loop_b": {"inherent_bytes": tot - loop_b_b, "frac": (tot - loop_b_b) / max((ms_step - avg_launch_s * 1e3) * 1e-3, 1e-9) / 1e9 / HBM_PEAK_GBS,
                                   "what": "loop A (keys in once, rotations out once: %.1f GB) + the per-block tails (relinearise, rescale, 22-product "
                                           "comparator: 2 N 8 B per limb-transform, keys once per launch, ct x ct operands once)" % (loop_a_b / 1e9)}}
        out = {
            "metric": "encrypted DB vectors matched/sec (HyDia indexScenario, CKKS N=2^15)",
            "value": n_total * args.steps / elapsed,
            "unit": "vectors/s",
            "n_gpus": world, "ranks_initialised": ranks_initialised, "collective_backend": backend,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": (n_total * args.steps / elapsed) / PUBLISHED_2P20_INDEX_VPS if (world == 1 and n_total == 1 << 20) else None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": workload_name(n_total, world, strong) + ": %d blocks of 16384 vectors on this GPU (%.0f GiB resident in "
                                   "HBM as %d-bit residues = %.0f GiB of 8-byte ciphertexts, %s), one query per step through indexScenario"
                                   % (G_local, db_gib, db_bits_timed, G_local * dim * 2 * nl * N * 8 / 2 ** 30,
                                      ("group-sequential layout, groups of %d blocks" % db_group_timed) if db_group_timed else "ciphertext-major layout"),
                       "db_vectors_total": n_total, "db": "random residues" if args.random_db else "real ciphertexts (GPU enroller)",
                       "ring": "N=2^15, 12 Q limbs (60+11x45 bit), 4 P limbs, dnum=3",
                       "matvec": ("baby-step / giant-step: %d hoisted rotations of the query, %d relinearised partial sums per block rotated by "
                                  "multiples of %d (pre-rotated diagonals; the split the auto rule picks for the %d blocks on this GPU)"
                                  % (n_rot - 1, dim // n_rot, n_rot, G_local)) if bsgs else
                                 "hoisted: %d hoisted rotations of the query, one relinearisation per block (the reference's form)" % (dim - 1),
                       "sharding": "row-blocks per GPU (image_matching_amd.sharding.DistDiagonalSender): query broadcast, loop A %s, "
                                   "independent mat-vec per rank, RCCL gather of result ciphertexts in global block order"
                                   % ("shared out over the ranks and all-gathered" if (loop_a and loop_a["mode"] == "split") else "recomputed by every rank")
                                   if multi else "one GPU",
                       "loop_a_across_ranks": loop_a,
                       "result_check": "decrypted index of the gathered result == planted matches (global indices)" if not args.random_db else "skipped (random DB)",
                       "result_correct": bool(correct), "setup_s": {"keygen": round(t_keygen, 2), "enroll": round(t_enroll, 2)},
                       "vs_baseline_note": "value / 10 864 vectors/s = the reference's published 2^20 index computation on a 48-thread Xeon "
                                           "(tools/figures/approach5.csv:12); null unless this run is that workload on one GPU",
                       "commit": git_head()},
            "roofline": {"kernel": "k_hydia_tensor24 + k_hydia_tensor (loop B: limbs 1-11 and limb 0, timed together)", "bound": "hbm",
                         "achieved": wire, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": wire / HBM_PEAK_GBS,
                         "bytes_per_launch": resident_bytes, "avg_launch_ms": avg_launch_s * 1e3, "launches": int(launches),
                         "traffic": traffic, "traffic_rate": (traffic / avg_launch_s / 1e9) if (traffic and launches) else None,
                         "traffic_meta": traffic_meta,
                         "algorithmic_frac": achieved / HBM_PEAK_GBS, "algorithmic_achieved": achieved, "algorithmic_bytes_per_launch": algo_bytes,
                         "vs_measured_stream_ceiling": (wire / stream_ceiling(db_bits_timed == 46)[0]) if stream_ceiling(db_bits_timed == 46)[0] else None,
                         "stream_ceiling": {"GBs": stream_ceiling(db_bits_timed == 46)[0], "source": stream_ceiling(db_bits_timed == 46)[1]},
                         "step": step_roofline,
                         "note": "achieved/frac = bytes RESIDENT in HBM that one loop-B pass has to move (database with 46- or 48-bit residues for the "
                                 "45/46-bit limbs, see config.workload, + rotated queries + accumulators at 8 bytes: every byte once) / mean pass duration (HIP events on "
                                 "the library's stream) / 8 TB/s: a utilisation, <= 1.  algorithmic_* = the same pass priced by SURVEY 8d at 8 bytes "
                                 "per residue (196608 B per DB vector): it exceeds `achieved` by the 46-bit storage and can pass 1.0 — a byte-saving "
                                 "figure, not a utilisation.  vs_measured_stream_ceiling: against stream_ceiling.GBs, what this GPU gives a read-once sequential "
                                 "stream in this access pattern (tools/ubench/stream_rate.hip, read from stream_ceiling.source; the guide's copy ceiling is 6.29).  traffic = PMC HBM bytes per launch "
                                 "(profiles/tensor_traffic.json), quoted only while the kernel source it was profiled on is unchanged.  step = the whole "
                                 "indexScenario: inherent bytes of every operation (byte ledger, op:* entries) / ms_per_step"},
        }
        if split_report is not None:
            mode = "local" if (loop_a and loop_a["mode"] == "local") else ("split" if getattr(sender, "rotation_split", False) else "replicated")
            out["comm_ms"], out["compute_ms"] = split_report["comm_ms"], split_report["compute_ms"]
            out["step_split"] = split_report
            out["loop_a_mode"] = mode
            out["model"] = model_step(n_total.bit_length() - 1, world, mode) if strong or world == 1 else None
            if out["model"]:
                out["model"]["measured_ms_per_step"] = round(ms_step, 3)
        if weak is not None:
            out["config"]["secondary"] = {"weak_scaling": weak}
        if ms_similarity is not None:
            out["config"]["secondary"] = {"computeSimilarity_ms_per_query": round(ms_similarity, 3),
                                          "computeSimilarity_vectors_per_s": round(n_local / ms_similarity * 1e3),
                                          "membershipScenario_ms_per_query": round(ms_membership, 3),
                                          "membershipScenario_result": membership_ok}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    del res, qc
    cc.close()
    if not correct:
        sys.exit(3)


def workload_name(n_total, world, strong):
    """the BASELINE.json configuration a run corresponds to"""
    lg = n_total.bit_length() - 1
    if world == 1:
        tag = {10: "BASELINE config 2 (2^10 DB, one GPU)", 14: "BASELINE config 3 (2^14 DB, one GPU)",
               20: "BASELINE headline (north_star: 2^20-vector / 512-dim database at 1 GPU)"}.get(lg, "")
        return "HyDia approach 5, 2^%d-vector x 512-dim encrypted DB on one MI355X%s" % (lg, " = " + tag if tag else "")
    if strong:
        tag = " = BASELINE config 5" if (lg, world) == (20, 8) else " = BASELINE config 4" if (lg, world) == (17, 4) else \
              " (BASELINE config 5's database on %d GPUs)" % world if lg == 20 else ""
        return "HyDia approach 5, 2^%d-vector x 512-dim encrypted DB sharded by row-blocks across %d x MI355X%s" % (lg, world, tag)
    return "HyDia approach 5, weak scaling: %d x MI355X with 2^%d vectors each (2^%.2f in all)" % (world, (n_total // world).bit_length() - 1, math.log2(n_total))


def kernel_sha():
    """identifies the loop-B kernel the committed PMC traffic figure belongs to: hash of the source text of k_hydia_tensor (from its
    template line to the end of its body) and of the packed-residue loader it streams the database with"""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "image_matching_amd", "csrc")
    text = open(os.path.join(src, "kernels.hip")).read()
    for start in ("template <int BPP, int NW, bool NT, bool PK>", "struct Acc24 {", "DEV DbWalk db_walk("):  # both loop-B kernels, the layout walk
        a = text.find(start)
        b = text.find("\n}\n", text.find("__global__", a) if "template" in start or "Acc24" in start else a)
        h.update(text[a:b].encode())
    hdr = open(os.path.join(src, "kernels.h")).read()
    for start in ("DEV ulonglong2 db_load2", "struct DbRaw<true>"):
        a = hdr.find(start)
        b = hdr.find("\n}", a)
        h.update(hdr[a:b].encode())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    main()
