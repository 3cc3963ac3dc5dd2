"""What does the data-parallel wrapper's bucket plumbing cost by itself (one rank, no collective)?  ms / step against the bucket count.
    python tools/ddp_overhead.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29612")
    dist.init_process_group("gloo", rank=0, world_size=1)
    from volta_amd.config import BertConfig
    from volta_amd.modeling import BertForVLPreTraining
    from volta_amd.optimization import AdamW, clip_grad_norm_
    from volta_amd.parallel import DistributedDataParallel
    from volta_amd.data import synthetic_batch, model_args
    cfg = BertConfig.from_json_file(os.path.join(ROOT, "config", "ctrl_vilbert_base.json"))
    torch.manual_seed(1234)
    model = BertForVLPreTraining(cfg).cuda()
    model.train()
    model.materialize()
    opt = AdamW(model.parameters(), lr=1e-4, overlap_with_forward=True)
    args = model_args(synthetic_batch(cfg, 256, 20, 36, seed=1234))

    def one(net):
        lm, img, nsp = net(*args)
        (lm + img + nsp).backward()
        clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        opt.zero_grad()

    def run(net, steps=12):
        for _ in range(3):
            one(net)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one(net)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, (t1 - t0) / steps * 1e3

    print("no wrapper: %.3f ms / step (host loop %.3f)" % run(model), flush=True)
    # which pool streams share a hardware queue with the compute stream / the weight-gradient side stream?
    import ctypes as C
    from volta_amd import _lib as L, streams as S
    own, side = S.engine_streams()
    cands, table = [], []
    for i in range(12):
        st = torch.cuda.Stream()
        cands.append(st)
        table.append((S.shares_queue(own, st), S.shares_queue(side, st)))
    print("GPU_MAX_HW_QUEUES=%s; pool stream i shares a hardware queue with (compute, side): %s" % (
        os.environ.get("GPU_MAX_HW_QUEUES"), " ".join("%d:%s%s" % (i, "C" if c else "-", "S" if s_ else "-") for i, (c, s_) in enumerate(table))), flush=True)
    for what, pick in (("a stream sharing the SIDE stream's queue", lambda c, s_: s_ and not c), ("a stream sharing the COMPUTE stream's queue", lambda c, s_: c and not s_),
                       ("a stream on a queue of its own", lambda c, s_: not c and not s_)):
        idx = [i for i, (c, s_) in enumerate(table) if pick(c, s_)]
        if not idx:
            print("%s: none among the candidates" % what)
            continue
        ddp = DistributedDataParallel(model, message_size=10000000)
        ddp.reducer.stream = cands[idx[0]]
        ddp.reducer._reduce = lambda ranges: None
        ms, host = run(ddp)
        print("reducer on %-46s (pool stream %d): %.3f ms / step" % (what, idx[0], ms), flush=True)
        model.__dict__["_ddp"] = None
    for what, kw in (("delay_allreduce (1 bucket)", dict(delay_allreduce=True)), ("message_size 1e8 (3 buckets)", dict(message_size=100000000)),
                     ("message_size 4e7", dict(message_size=40000000)), ("message_size 1e7 (apex default)", dict(message_size=10000000))):
        ddp = DistributedDataParallel(model, **kw)
        ddp.reducer._reduce = lambda ranges: None
        ms, host = run(ddp)
        eng = model._last[0]
        print("%-34s: %.3f ms / step (host loop %.3f), %d buckets" % (what, ms, host, len(ddp._plan(eng))), flush=True)
        # the same buckets, but the communication stream does not wait for the weight-gradient side stream
        ddp.reducer.reduce = lambda ranges, join=None: None
        ms, host = run(ddp)
        print("%-34s: %.3f ms / step with reduce() a no-op (list cut only)" % ("", ms), flush=True)
        model.__dict__["_ddp"] = None
    dist.destroy_process_group()


main()
