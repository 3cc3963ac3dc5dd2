"""What would RCCL's channel kernels cost beside the backward pass, and what would they get?    One GPU cannot run
RCCL with more than one rank, so the bucket collectives of volta_amd.parallel.DistributedDataParallel are replaced by a STAND-IN with the
same footprint: `--wgs` workgroups of 256 threads that stream the bucket to a second buffer and stay resident for the time the bucket would
take on the links (`--gbps` algorithm bandwidth), launched on the communication stream exactly where the collective would be.
Prints, per configuration: ms / step, and per bucket the time the stand-in waited for CUs (stream time minus in-kernel span).
    python tools/comm_footprint.py [--wgs 32] [--gbps 300] [--reserve 0,16,32]"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wgs", type=int, default=32)
    ap.add_argument("--gbps", type=float, default=300.0)
    ap.add_argument("--reserve", default="0,16,32")
    ap.add_argument("--steps", type=int, default=12)
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("gloo", rank=0, world_size=1)
    from volta_amd.config import BertConfig
    from volta_amd.modeling import BertForVLPreTraining
    from volta_amd.optimization import AdamW, clip_grad_norm_
    from volta_amd.parallel import DistributedDataParallel
    from volta_amd.data import synthetic_batch, model_args
    from volta_amd import _lib as L
    cfg = BertConfig.from_json_file(os.path.join(ROOT, "config", "ctrl_vilbert_base.json"))
    torch.manual_seed(1234)
    model = BertForVLPreTraining(cfg).cuda()
    model.train()
    arena = model.materialize()
    ddp = DistributedDataParallel(model, message_size=10000000)
    red = ddp.reducer
    opt = AdamW(model.parameters(), lr=1e-4, overlap_with_forward=True)
    args = model_args(synthetic_batch(cfg, 256, 20, 36, seed=1234))
    sink = torch.empty_like(arena.grad)
    stamps = torch.zeros(64, 2, dtype=torch.int64, device="cuda")
    state = dict(mode="none", k=0, events=[])

    def standin(ranges):
        for lo, hi in ranges:
            if state["mode"] == "none":
                continue
            k = state["k"]
            state["k"] += 1
            nbytes = (hi - lo) * 4
            usec = int(nbytes * 2 * 7 / 8 / (a.gbps * 1e3))              # ring / direct all-reduce at N = 8: 2 (N-1)/N x bytes at the algorithm bandwidth
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(L.lib.vk_comm_standin(C.c_void_p(arena.grad.data_ptr() + 4 * lo), C.c_void_p(sink.data_ptr() + 4 * lo), nbytes, a.wgs, usec,
                                          C.c_void_p(stamps[k].data_ptr()), L.stream_ptr()))
            e1.record()
            state["events"].append((k, nbytes, usec, e0, e1))

    red._reduce = standin

    def run(net, steps):
        for _ in range(3):
            one(net)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one(net)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    def one(net):
        state["k"], state["events"] = 0, []
        stamps[:, 0] = (1 << 62)
        stamps[:, 1] = 0
        lm, img, nsp = net(*args)
        (lm + img + nsp).backward()
        clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        opt.zero_grad()

    print("workgroups per stand-in %d x 256 threads, resident for bucket bytes x 1.75 / %.0f GB/s" % (a.wgs, a.gbps))
    model.__dict__["_ddp"] = None
    print("no wrapper                         : %.3f ms / step" % run(model, a.steps), flush=True)
    model.__dict__["_ddp"] = ddp
    print("wrapper, buckets cut, no collective: %.3f ms / step" % run(ddp, a.steps), flush=True)
    for reserve in [int(x) for x in a.reserve.split(",")]:
        L.lib.vk_gemm_reserve_cus(reserve)
        state["mode"] = "none"
        base = run(ddp, a.steps)
        state["mode"] = "standin"
        ms = run(ddp, a.steps)
        torch.cuda.synchronize()
        waits, spans, ideal = [], [], []
        for k, nbytes, usec, e0, e1 in state["events"]:
            span = float(stamps[k, 1] - stamps[k, 0]) / 100.0
            waits.append(e0.elapsed_time(e1) * 1e3 - span)
            spans.append(span)
            ideal.append(usec)
        print("reserved CUs %3d: %.3f ms / step without, %.3f with the stand-in (%d buckets) | stand-in waited for CUs: median %.0f us, max %.0f us, sum %.0f us; "
              "resident %.0f us in all (the links' time: %.0f us)" % (reserve, base, ms, len(waits), sorted(waits)[len(waits) // 2], max(waits), sum(waits), sum(spans), sum(ideal)), flush=True)
    L.lib.vk_gemm_reserve_cus(0)
    dist.destroy_process_group()


main()
