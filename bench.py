#!/usr/bin/env python3
"""Benchmark of the hot path: one full pre-training step (forward + backward + grad-norm clip + AdamW
[+ gradient all-reduce]) of BertForVLPreTraining on synthetic ConceptualCaptions-shaped batches.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched through torch.distributed.run)

Prints ONE JSON line (rank 0): image-text pairs/s over all GPUs, the MFMA roofline of the step and of the
GEMM kernel family (timed live with HIP events on the launch stream), and a CPU baseline (the fp32 oracle
of oracle/volta_ref.py on the host cores, bounded sample)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the HIP runtime starts: one hardware queue per engine stream (volta_amd/streams.py)

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# matmul FLOPs per image-text pair, forward + backward, measured on the reference graph
GFLOP_PER_PAIR = {("ctrl_vilbert_base", 20, 36): 37.876, ("ctrl_vilbert_base", 38, 36): 54.420, ("ctrl_lxmert", 20, 36): 35.415,
                  ("ctrl_uniter_base", 20, 36): 32.937, ("ctrl_visualbert_base", 20, 36): 32.937, ("ctrl_vl-bert_base", 20, 100): 69.184}
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_FP8_TFLOPS = 5000.0         # dense fp8 MFMA peak: the yardstick BASELINE.json configs[4] names for --dtype fp8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="ctrl_vilbert_base")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--seq-len", type=int, default=20)
    ap.add_argument("--regions", type=int, default=36)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"], help="arithmetic of the projection GEMMs (fp8: forward projections on the MX-scaled fp8 MFMA, bf16 elsewhere)")
    ap.add_argument("--no-opt-overlap", action="store_true", help="AdamW as one launch on the compute stream instead of range by range under the next forward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--serial", action="store_true", help="run the side-stream blocks (weight gradients) inline: per-kernel profiles without concurrency")
    ap.add_argument("--no-dropout", action="store_true", help="DIAGNOSTIC, not the BASELINE workload (the line says so): every dropout probability 0 -- what the counter-based mask generation costs the step")
    ap.add_argument("--no-optimizer", action="store_true", help="DIAGNOSTIC, not the BASELINE workload (the line says so): forward + backward only -- what clip + AdamW add to the step beyond what the next forward hides")
    ap.add_argument("--dump-ops", default=None, help="write the per-op timing table of the profiled steps to this file")
    return ap.parse_args()


def gemm_flops(eng, plan):
    """Algorithmic FLOPs (2 M N K) of every GEMM op in a plan; device-side row counts are read back."""
    from volta_amd import _lib as L
    out, bytes_out = {}, {}
    for i, (kind, layout, epi, nprob, arr, _, _) in enumerate(plan.ops):
        if kind not in (L.OP_GEMM, L.OP_GEMM_FP8, L.OP_GEMM_CHAIN):
            continue
        layout &= 0xFF                     # the op carries a tile geometry above the layout
        fl, by = 0.0, 0.0
        if kind == L.OP_GEMM_CHAIN:        # producers (a) then consumers (b) of one persistent launch; epilogues packed in `epi`
            probs = [(arr[j], epi & 0xFF) for j in range(nprob & 0xFF)] + [(plan.ops[i][5][j], epi >> 8) for j in range(nprob >> 8)]
        else:
            probs = [((arr[j].p if kind == L.OP_GEMM_FP8 else arr[j]), epi) for j in range(nprob)]
        for q, epi in probs:
            M, K = q.M, q.K
            if q.dyn:
                n = int(_dev_int(q.dyn, eng).item())
                if layout == L.TN:
                    K = min(K, n)
                else:
                    M = min(M, n)
            fl += 2.0 * M * q.N * K
            eo = 1 if kind == L.OP_GEMM_FP8 else 2                                     # operand bytes per element
            ec = 4 if epi in (L.EPI_F32, L.EPI_F32_ACC) else 2                         # output bytes per element
            by += (M * K + q.N * K) * eo + M * q.N * ec * (2 if epi == L.EPI_GELU else 1)     # every operand read once, every output written once
            if epi in (L.EPI_MULR, L.EPI_ADDR):
                by += M * q.N * 2                                                      # the multiplier / residual operand
            if epi == L.EPI_F32_ACC:
                by += M * q.N * 4
        out[i] = fl
        bytes_out[i] = by
    gemm_flops.compulsory_bytes = bytes_out
    return out


def _dev_int(addr, eng):
    for name in ("n_t", "n_v"):
        if eng.bufs[name].data_ptr() == addr:
            return eng.bufs[name].cpu()
    raise KeyError(addr)


def source_fingerprint():
    """sha1 over the HIP / C++ sources of the library: ties a committed PMC summary to the kernels it measured."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "volta_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")) and name not in ("records.cpp", "wordpiece.cpp", "unicode_tables.h"):      # host-side readers / tokenizer: launch nothing
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def usable_cpus():
    """Host cores this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box shows all 256 host
    cores to a container that is granted a 16-CPU share; 256 torch threads on 16 CPUs run ~50x slower than 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                       # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = fh.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:      # cgroup v1
                q, per = float(fq.read()), float(fp.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    elif n > 64:
        n = 16        # a whole host's cores visible and no quota readable: assume the pool's documented 16-CPU share per GPU
    cap = os.environ.get("VK_BENCH_CPU_THREADS")
    if cap:
        n = min(n, int(cap))
    return max(1, n)


def cpu_baseline(cfg_name, T, R, budget_s=75.0):
    """fp32 oracle (the reference's arithmetic in stock torch CPU ops): forward + backward + clip + AdamW, B = 32, on every
    host core the process may use (count printed), SURVEY.md 8(d) protocol: 3 warm-up + 5 timed steps.  The time budget
    only cuts the sample short on a slow host (the line then says how many steps were timed)."""
    from oracle import volta_ref as Rf
    ncpu = usable_cpus()
    torch.set_num_threads(max(1, ncpu))
    print("[bench] cpu baseline on %d threads (os.cpu_count() = %s, usable = %d)" % (torch.get_num_threads(), os.cpu_count(), ncpu), file=sys.stderr, flush=True)
    cfg = Rf.RefConfig.from_json_file(os.path.join(ROOT, "config", cfg_name + ".json"))
    sd = Rf.make_weights(cfg, seed=1, std=0.02)
    aliases = Rf.param_aliases(cfg)
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if k not in aliases}
    full = dict(leaves)
    for a, t in aliases.items():
        full[a] = leaves[t]
    m = {k: torch.zeros_like(v) for k, v in leaves.items()}
    v2 = {k: torch.zeros_like(v) for k, v in leaves.items()}
    B, n_warm, n_timed = 32, 3, 5
    batch = Rf.synthetic_batch(cfg, B, T, R, seed=3)
    times = []
    t_start = time.time()
    for step in range(1, n_warm + n_timed + 1):
        t0 = time.time()
        lm, img, nsp = Rf.forward_from_batch(full, cfg, batch, train=True)
        for p in leaves.values():
            p.grad = None
        (lm + img + nsp).sum().backward()
        grads = [p.grad for p in leaves.values()]
        Rf.clip_grad_norm(grads, 5.0)
        with torch.no_grad():
            for k, p in leaves.items():
                Rf.adamw_step(p, p.grad, m[k], v2[k], step, 1e-4, 0.9, 0.999, 1e-6, 0.01 if Rf.decays(k) else 0.0)
        times.append(time.time() - t0)
        print("[bench] cpu step %d: %.2f s" % (step, times[-1]), file=sys.stderr, flush=True)
        if time.time() - t_start > budget_s and len(times) >= 2:
            break
    n_w = min(n_warm, len(times) - 1)
    timed = times[n_w:]
    return dict(value=B * len(timed) / sum(timed), unit="image-text pairs/s", cores=torch.get_num_threads(), kind="port",
                sample="%d timed steps (after %d warm-up) of fwd+bwd+clip+AdamW, fp32 oracle, B=32, T=%d, R=%d, %s" % (len(timed), n_w, T, R, cfg_name))


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks through torch.distributed.run as a CHILD process (this
    process has not touched the GPU yet -- never re-exec one that has) and hand its exit code back."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def result_line(a, world, elapsed, t_issue, extra):
    ms = elapsed * 1e3 / a.steps
    pairs_s = a.batch * world * a.steps / elapsed
    out = {"metric": "image-text pairs/sec, %s pretrain step" % a.config, "value": pairs_s, "unit": "image-text pairs/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": "%s pretrain step (%s%s), per-GPU batch %d, T=%d, %d regions (+1 global), objective 1, dropout %s"
                      % (a.config, "fwd+bwd ONLY: diagnostic run, not the BASELINE workload" if a.no_optimizer else "fwd+bwd+clip+AdamW", "+allreduce" if world > 1 else "",
                         a.batch, a.seq_len, a.regions, "OFF (diagnostic run, not the BASELINE workload)" if a.no_dropout else "on"),
                      "global_batch": a.batch * world, "seq_len": a.seq_len, "regions": a.regions, "parallelism": "dp%d" % world},
           "host_issue_ms_per_step": t_issue * 1e3 / a.steps}
    out.update(extra)
    return out, ms, pairs_s


def dry_run(a, world, rank):
    """VK_BENCH_DRY_RUN=1: the launcher / rendezvous / barrier / max-over-ranks / one-JSON-line plumbing of the N-rank
    benchmark on CPU over gloo, with a small all-reduce standing in for the step.  A test of the harness, not a measurement."""
    if world > 1:
        dist.init_process_group("gloo")
    buf = torch.ones(1024)

    def step():
        if world > 1:
            dist.all_reduce(buf)
            buf.mul_(1.0 / world)

    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_issue = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    out, _, _ = result_line(a, world, elapsed, t_issue, {"dry_run": "harness rehearsal on CPU over gloo: no model step ran, the value means nothing"})
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    if os.environ.get("VK_BENCH_DRY_RUN") == "1":
        return dry_run(a, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU path")
    # VK_BENCH_SHARED_GPU=1 is a REHEARSAL mode for the multi-rank code path on a one-GPU box: every rank uses cuda:0 and the
    # collectives go through gloo (RCCL refuses two ranks on one device); its timings mean nothing and the output says so.
    shared = os.environ.get("VK_BENCH_SHARED_GPU") == "1"
    if shared:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from volta_amd.config import BertConfig
    from volta_amd.modeling import BertForVLPreTraining
    from volta_amd.optimization import AdamW, WarmupLinearSchedule, clip_grad_norm_
    from volta_amd.parallel import DistributedDataParallel
    from volta_amd.data import synthetic_batch, model_args
    from volta_amd import _lib as L
    if a.serial:
        L.lib.vk_side_enable(0)

    cfg = BertConfig.from_json_file(os.path.join(ROOT, "config", a.config + ".json"))
    if a.no_dropout:
        for k in ("hidden_dropout_prob", "attention_probs_dropout_prob", "v_hidden_dropout_prob", "v_attention_probs_dropout_prob"):
            setattr(cfg, k, 0.0)
    torch.manual_seed(1234)
    model = BertForVLPreTraining(cfg).cuda()
    model.train()
    model.materialize()
    model.set_projection_dtype(a.dtype)
    peak = PEAK_FP8_TFLOPS if a.dtype == "fp8" else PEAK_BF16_TFLOPS
    net = DistributedDataParallel(model) if world > 1 else model
    no_decay = ("bias", "LayerNorm.bias", "LayerNorm.weight")
    groups = [{"params": [p], "lr": 1e-4, "weight_decay": 0.0 if any(nd in n for nd in no_decay) else 0.01}
              for n, p in model.named_parameters()]                       # one group per parameter (train_concap.py:213-224)
    opt = AdamW(groups, lr=1e-4, eps=1e-6, betas=(0.9, 0.999), overlap_with_forward=not a.no_opt_overlap)
    sched = WarmupLinearSchedule(opt, warmup_steps=100, t_total=100000)
    batch = synthetic_batch(cfg, a.batch, a.seq_len, a.regions, seed=1234 + rank)
    args = model_args(batch)

    def step():
        lm, img, nsp = net(*args)
        loss = lm + img + nsp
        loss.backward()
        if not a.no_optimizer:
            clip_grad_norm_(model.parameters(), 5.0)
            opt.step()
            sched.step()
        opt.zero_grad()
        return lm, img, nsp

    for _ in range(a.warmup):
        losses = step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = step()
    t_issue = time.perf_counter() - t0       # host time to ISSUE the steps (the launches are asynchronous)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    # Host WORK per step: the same step issued into an EMPTY queue (synchronise first), on every rank alike.  `host_issue_ms_per_step` above
    # is the wall time of the issuing loop, which the bounded HIP queues throttle to the GPU's pace once they are full -- it approaches
    # ms_per_step by construction and says nothing about host cost; this figure does.
    host_work = []
    for _ in range(0 if a.no_kernel_timing else 3):       # (profiling runs count launches per step: no extra steps there)
        fence()
        th = time.perf_counter()
        step()
        host_work.append(time.perf_counter() - th)
    fence()
    gflop = GFLOP_PER_PAIR.get((a.config, a.seq_len, a.regions))
    out, ms, pairs_s = result_line(a, world, elapsed, t_issue, {
        "host_work_ms_per_step": min(host_work) * 1e3 if host_work else None,
        "host_note": "host_work = one step issued into an empty queue (the host's own cost); host_issue = wall time of the issuing loop, throttled "
                     "by queue back-pressure to the GPU's pace",
        **({"rehearsal": "all ranks share cuda:0 over gloo: timings are meaningless"} if shared else {}),
        "losses_last_step": [float(x.detach()) for x in losses],
        # 0 = no guarded tile of a chained GEMM launch ever gave up waiting for its row block (engine.soft_error(); a wrong schedule would show here, not as a hang)
        "handoff_errors": model._last[0].soft_error()})
    if rank == 0:
        print("[bench] timed region done: %.3f ms/step, %.1f pairs/s" % (ms, pairs_s), file=sys.stderr, flush=True)
        if gflop is not None:
            ach = pairs_s / world * gflop / 1e3            # per-GPU TFLOP/s
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                               "traffic": None, "scope": "whole step: pairs/s x %.3f GFLOP/pair (reference-graph matmul FLOPs, BASELINE.md)" % gflop}
            if a.dtype == "fp8":
                out["roofline"]["peak_note"] = ("dense fp8 MFMA peak; only the forward Q|K|V / FFN projections run on the e4m3 MFMA, attention output, "
                                                "heads and the whole backward stay bf16 (peak 2500)")
        if world > 1 and "roofline" in out:
            out["roofline"]["note"] = ("N > 1: this is the whole-step figure per GPU (pairs/s / N x GFLOP/pair); the GEMM-family roofline with its live "
                                       "launch timing, `traffic` (PMC) and `cpu_baseline` are measured at N = 1 only -- profiled extra steps would "
                                       "desynchronise the ranks' collectives")
        if not a.no_kernel_timing and world == 1:      # extra profiled steps would desynchronise the ranks' collectives
            eng = model._last[0]
            eng.fwd.enable_timing(True)
            eng.bwd.enable_timing(True)
            nprof = 3
            for _ in range(nprof):
                step()
            torch.cuda.synchronize()
            fl = dict(("f%d" % i, v) for i, v in gemm_flops(eng, eng.fwd).items())
            comp_bytes = sum(gemm_flops.compulsory_bytes.values())
            fl.update(("b%d" % i, v) for i, v in gemm_flops(eng, eng.bwd).items())
            comp_bytes += sum(gemm_flops.compulsory_bytes.values())
            tms = dict(("f%d" % i, eng.fwd.timing[i] / nprof) for i in range(len(eng.fwd.ops)))
            tms.update(("b%d" % i, eng.bwd.timing[i] / nprof) for i in range(len(eng.bwd.ops)))
            gemm_ms = sum(tms[k] for k in fl)
            gemm_fl = sum(fl.values())
            kinds = {}
            for plan, tag in ((eng.fwd, "f"), (eng.bwd, "b")):
                for i, op in enumerate(plan.ops):
                    name = {L.OP_GEMM: "gemm", L.OP_GEMM_CHAIN: "gemm", L.OP_GEMM_FP8: "gemm_fp8", L.OP_LN_FWD: "ln_fwd", L.OP_LN_BWD: "ln_bwd", L.OP_ATTN_FWD: "attn_fwd", L.OP_ATTN_BWD: "attn_bwd"}.get(op[0], "other")
                    kinds[name] = kinds.get(name, 0.0) + tms["%s%d" % (tag, i)]
            if a.dump_ops:
                with open(a.dump_ops, "w") as fh:
                    for plan, tag in ((eng.fwd, "f"), (eng.bwd, "b")):
                        for i, op in enumerate(plan.ops):
                            key = "%s%d" % (tag, i)
                            if op[0] == L.OP_GEMM_CHAIN:
                                pr = [op[4][j] for j in range(op[3] & 0xFF)] + [op[5][j] for j in range(op[3] >> 8)]
                                fh.write("%s gemm chain layout=%d epi=%d->%d [%s] %.1f us %.0f TF/s\n" % (key, op[1], op[2] & 0xFF, op[2] >> 8, ";".join("%dx%dx%d" % (q.M, q.N, q.K) for q in pr),
                                                                                                       tms[key] * 1e3, fl[key] / (tms[key] * 1e-3) / 1e12))
                            elif op[0] in (L.OP_GEMM, L.OP_GEMM_FP8):
                                arr, nprob = op[4], op[3]
                                pr = [(arr[j].p if op[0] == L.OP_GEMM_FP8 else arr[j]) for j in range(min(nprob, 4))]
                                shapes = ("fp8 " if op[0] == L.OP_GEMM_FP8 else "") + ";".join("%dx%dx%d" % (q.M, q.N, q.K) for q in pr)
                                fh.write("%s gemm layout=%d epi=%d nprob=%d [%s] %.1f us %.0f TF/s\n" % (key, op[1] & 0xFF, op[2], nprob, shapes, tms[key] * 1e3, fl[key] / (tms[key] * 1e-3) / 1e12))
                            else:
                                fh.write("%s kind=%d %.1f us\n" % (key, op[0], tms[key] * 1e3))
            eng.fwd.enable_timing(False)
            eng.bwd.enable_timing(False)
            ach = gemm_fl / (gemm_ms * 1e-3) / 1e12
            # The contract's `roofline` object describes the DOMINANT KERNEL (the bf16 MFMA GEMM family, ~65 % of the
            # step's kernel time): algorithmic FLOPs per launch / average launch duration, both measured live.  The
            # whole-step figure (pairs/s x reference-graph FLOPs per pair) is kept beside it.
            whole = out["roofline"]
            out["roofline"] = {
                "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": None,
                "kernel": "vk::gemm256k_kernel / gemm256p_kernel / gemm_kernel (bf16 MFMA 16x16x32 GEMM family, all layouts and epilogues)"
                          + (" + vk::gemm_fp8_kernel (e4m3 MFMA 16x16x128, forward projections); priced against the bf16 peak" if a.dtype == "fp8" else ""),
                "launches_per_step": len(fl), "algorithmic_gflop_per_launch": gemm_fl / 1e9 / len(fl), "avg_launch_us": gemm_ms * 1e3 / len(fl),
                "compulsory_mbytes_per_launch": comp_bytes / 1e6 / len(fl),     # every operand read once + every output written once
                "ms_per_step": gemm_ms, "timing": "HIP events on the launch stream around every launch, %d profiled steps, serial schedule" % nprof,
                "sustained_mfma_note": "a register-only MFMA loop sustains ~1.75-2.0 PFLOP/s on this part (tools/bench_gemm.py peak); `peak` is the datasheet 2.5",
                "whole_step": {"achieved": whole["achieved"], "frac": whole["frac"], "scope": whole["scope"]}}
            out["kernel_ms_per_step"] = {k: round(v, 3) for k, v in sorted(kinds.items(), key=lambda kv: -kv[1])}
            # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the process, so the figure is the
            # one of the committed rocprofv3 --pmc passes over this same command (profiles/r04_pmc_hbm_traffic.md)
            # -- and only while that pass describes THIS library: the summary carries a fingerprint of the kernel sources it profiled
            pmc = os.path.join(ROOT, "profiles", "r04_pmc_summary.json")
            if os.path.exists(pmc) and a.config == "ctrl_vilbert_base" and a.batch == 256 and a.dtype == "bf16":
                pm = json.load(open(pmc))
                if pm.get("source_fingerprint") == source_fingerprint():
                    out["roofline"]["traffic"] = pm["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_note"] = ("HBM bytes per GEMM launch (avg of %d launches/step), FETCH_SIZE x2 + WRITE_SIZE from profiles/r04_pmc_summary.json "
                                                       "(rocprofv3 --pmc passes over this command; kernel sources unchanged since)" % pm["launches_per_step"])
                else:
                    out["roofline"]["traffic_note"] = "profiles/r04_pmc_summary.json was collected on different kernel sources: traffic withheld (re-run tools/collect_profiles.sh)"
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(a.config, a.seq_len, a.regions)
            except Exception as e:      # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
