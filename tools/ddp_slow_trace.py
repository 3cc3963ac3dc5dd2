"""Find a data-parallel wrapper instance whose step is slow (tools/ddp_overhead.py: 22-24 ms instead of 16.9) and leave its steps at the end of
the process, so that a kernel trace of this program ends in slow steps:  rocprofv3 --kernel-trace ... -- python3 tools/ddp_slow_trace.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29613")
    dist.init_process_group("gloo", rank=0, world_size=1)
    from volta_amd.config import BertConfig
    from volta_amd.modeling import BertForVLPreTraining
    from volta_amd.optimization import AdamW, clip_grad_norm_
    from volta_amd.parallel import DistributedDataParallel
    from volta_amd.data import synthetic_batch, model_args
    from volta_amd import streams as S
    cfg = BertConfig.from_json_file(os.path.join(ROOT, "config", "ctrl_vilbert_base.json"))
    torch.manual_seed(1234)
    model = BertForVLPreTraining(cfg).cuda()
    model.train()
    model.materialize()
    opt = AdamW(model.parameters(), lr=1e-4, overlap_with_forward=os.environ.get("VK_NO_OPT_OVERLAP") != "1")
    args = model_args(synthetic_batch(cfg, 256, 20, 36, seed=1234))

    def one(net):
        lm, img, nsp = net(*args)
        (lm + img + nsp).backward()
        clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        opt.zero_grad()

    def run(net, steps=8):
        for _ in range(3):
            one(net)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one(net)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    print("no wrapper: %.3f ms / step" % run(model), flush=True)
    want_slow = os.environ.get("VK_WANT", "slow") == "slow"
    if os.environ.get("VK_SWEEP"):
        # distribution over wrapper instances (each takes another pool stream for its reducer): min / median / max ms per step
        res = []
        for attempt in range(int(os.environ["VK_SWEEP"])):
            ddp = DistributedDataParallel(model, message_size=10000000)
            ddp.reducer._reduce = lambda ranges: None
            res.append(run(ddp, 6))
            own, side = S.engine_streams()
            cs = ddp.reducer.stream
            ostream = opt._fused.get("stream")
            print("  instance %d: %.2f ms / step | comm stream %#x | launches on compute with the comm stream active / idle: %.0f / %.0f us; on side: %.0f / %.0f; on the optimizer stream: %.0f / %.0f; "
                  "launches on the comm stream with compute active: %.0f / %.0f, side active: %.0f / %.0f, optimizer active: %.0f / %.0f" % (
                (attempt, res[-1], cs.cuda_stream) + S.active_cost(own, cs) + S.active_cost(side, cs) + S.active_cost(ostream, cs)
                + S.active_cost(cs, own) + S.active_cost(cs, side) + S.active_cost(cs, ostream)), flush=True)
            model.__dict__["_ddp"] = None
        res_s = sorted(res)
        print("wrapper instances: %s | min %.3f median %.3f max %.3f ms / step" % (" ".join("%.2f" % r for r in res), res_s[0], res_s[len(res_s) // 2], res_s[-1]), flush=True)
        dist.destroy_process_group()
        return
    for attempt in range(12):
        ddp = DistributedDataParallel(model, message_size=100000000 if attempt % 2 == 0 else 10000000)
        ddp.reducer._reduce = lambda ranges: None
        ms = run(ddp)
        own, side = S.engine_streams()
        ostream = opt._fused.get("stream")
        cs = ddp.reducer.stream
        print("attempt %d (message_size %g): %.3f ms / step | comm stream shares a queue with compute %s, side %s, optimizer %s; optimizer with compute %s, side %s" % (
            attempt, ddp.message_size, ms, S.shares_queue(own, cs), S.shares_queue(side, cs), S.shares_queue(ostream, cs) if ostream else None,
            S.shares_queue(own, ostream) if ostream else None, S.shares_queue(side, ostream) if ostream else None), flush=True)
        if (ms > 19.0) == want_slow:
            print("tracing this one", flush=True)
            run(ddp, 4)
            break
        model.__dict__["_ddp"] = None
    dist.destroy_process_group()


main()
