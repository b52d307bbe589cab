"""Where does the multi-rank step spend its extra time?  One-rank RCCL group, random DB/keys, per-stage host timings."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import image_matching_amd as im

mode = sys.argv[1] if len(sys.argv) > 1 else "dist"
if mode == "torchcuda":
    torch.cuda.set_device(0)
    _t = torch.zeros(1024, device="cuda")
    torch.cuda.synchronize()
elif mode != "plain":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
cc = im.Context()
n = 1 << 20
cc.fill_eval_keys_random(1); cc.db_fill_random(n, 2)
rng = np.random.default_rng(0)
q = np.stack([rng.integers(0, int(m), size=(2, cc.N), dtype=np.uint64) for m in cc.moduli[:cc.nQ]], axis=1)
gq = cc.import_ct(q, cc.delta)
snd = im.DiagonalSender(cc, n)
buf = lst = None
for it in range(6):
    cc.sync(); 
    if mode not in ("plain",): torch.cuda.synchronize()
    t0 = time.time()
    res = snd.indexScenario(gq); t1 = time.time()
    if mode in ("plain", "torchcuda"):
        cc.sync(); t2 = t3 = time.time()
    else:
        cnt, npoly, nl, _ = res.shape()
        if buf is None:
            buf = torch.empty(cnt * npoly * nl * cc.N, dtype=torch.int64, device="cuda")
            lst = [torch.empty_like(buf)]
        res.copy_to_device(buf.data_ptr()); t2 = time.time()
        if mode == "dist":
            dist.gather(buf, lst, dst=0)
        torch.cuda.synchronize(); t3 = time.time()
    print("%s it%d: enqueue %.2f ms, to end of query %.2f ms, gather+sync %.2f ms, total %.2f" % (mode, it, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3), flush=True)
if mode not in ("plain", "torchcuda"):
    dist.destroy_process_group()
cc.close()
