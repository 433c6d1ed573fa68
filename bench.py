#!/usr/bin/env python
"""Benchmark of the S2VT hot path on MI355X (BASELINE.json metric: training frames/s + greedy captions/s).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full optimisation step of train.py:116-127 (zero_grad, forward, MaskCriterion, backward,
gradient all-reduce when N > 1, Adam) on synthetic [B, 80, 4096] fp32 features with seeded random-init
weights; the default workload is BASELINE configs[1] (B=64 per GPU, H=E=1000, V=12000, fp32).  Scaling is
weak: every rank keeps B per GPU.  Rank 0 prints ONE JSON line.  Inputs are resident in HBM before the timed
region.  Beside the GPU number the line carries
  * "roofline": the dominant kernel family of the step, timed live with HIP events on the launch stream
    (s2vt_prof_*), priced with the algorithmic flops/bytes of SURVEY.md §8(d);
  * "roofline_lstm_step": the fused LSTM timestep (north_star's target kernel) against the HBM roofline;
  * "cpu_baseline": the reference-shaped CPU model (oracle/, nn.LSTM/nn.Linear -> oneDNN, what the
    reference's CPU run executes) timed on this host's cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA = 157.3 TFLOP/s
MFMA_BF16_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: bf16 MFMA ~2.5 PFLOP/s dense


def gemm_flops_train(B, L, F, H, E, V, dfeats=False):
    """Algorithmic FLOPs of every batched GEMM launch of one train step (forward + backward), structural
    zeros excluded; the recurrent h·W_hh products live in the step kernels, not here."""
    T, R = 2 * L - 1, (L - 1) * B
    f = 0
    f += 2 * B * L * H * F                 # x1 = feats W_f^T
    f += 2 * L * B * 4 * H * H             # gx1
    f += 2 * T * B * 4 * H * H             # gx2 (vid_out half)
    f += 2 * R * 4 * H * E                 # gx2 (embed half)
    f += 2 * R * V * H                     # logits
    f += 2 * 2 * R * V * H                 # dh2dec, dW_o
    f += 2 * 4 * H * H * (T - 1) * B       # dW_hh2
    f += 2 * 4 * H * H * T * B             # dW_ih2[:, E:]
    f += 2 * 4 * H * E * R                 # dW_ih2[:, :E]
    f += 2 * T * B * H * 4 * H             # dh1
    f += 2 * R * E * 4 * H                 # d(embedded words)
    f += 2 * 4 * H * H * (T - 1) * B       # dW_hh1
    f += 2 * 4 * H * H * L * B             # dW_ih1
    f += 2 * L * B * H * 4 * H             # dx1
    f += 2 * H * F * L * B                 # dW_f
    if dfeats:
        f += 2 * L * B * F * H
    return f


def step_bytes_fwd(B, H, I, s=4, train=True):
    """ALGORITHMIC bytes of one fused LSTM timestep, SURVEY.md §8(d):
    W_ih+W_hh, biases, x_t and h_{t-1}, c_{t-1}, h_t, c_t (+ gate stash in training)."""
    b = s * 4 * H * (I + H) + 4 * 8 * H + s * B * (I + H) + 4 * B * H + s * B * H + 4 * B * H
    if train:
        b += s * B * 4 * H
    return b


_T0 = time.perf_counter()


def log(msg):
    """progress on stderr (stdout carries only the JSON line)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def usable_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota (on a shared host
    os.cpu_count() reports every core of the machine and oversubscribes the box's share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, int(os.environ.get("S2VT_CPU_THREADS", "16"))))


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: this process has not touched the GPU; it starts the N ranks as
    CHILD processes through torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) and relays rank 0's JSON line
    (the children inherit stdout).  Never an exec: a process that has initialised the GPU must not be replaced."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def launcher_selftest(args):
    """CPU rehearsal of the multi-rank plumbing (tests/test_bench_launcher.py): every rank joins a gloo group, the
    world size must equal --gpus, one all-reduce must see every rank, rank 0 prints the line."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    world = dist.get_world_size()
    if world != args.gpus:
        raise SystemExit("--gpus %d but the process group has %d ranks" % (args.gpus, world))
    t = torch.ones(1)
    dist.all_reduce(t)
    if int(t.item()) != world:
        raise SystemExit("all-reduce saw %d of %d ranks" % (int(t.item()), world))
    # the fixed-global-batch record's sharding (global 1024 -> 1024 / N rows per GPU): every rank reports its rows, rank 0 checks that
    # the shards tile the batch exactly once
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import s2vt_video_caption_amd  # noqa: F401
    from s2vt_video_caption_amd import dp as _dp
    per, lo, hi = _dp.fixed_global_shard(1024, dist.get_rank(), world)
    rows = [None] * world
    dist.all_gather_object(rows, (lo, hi))
    if sorted(rows) != [(r * per, (r + 1) * per) for r in range(world)] or per * world != 1024:
        raise SystemExit("fixed-global-batch shards do not tile the batch: %s" % rows)
    if dist.get_rank() == 0:
        print(json.dumps({"selftest": "launcher", "n_gpus": world, "ranks": world, "backend": "gloo",
                          "fixed_global": {"global_batch": 1024, "per_gpu_batch": per, "rows_by_rank": rows}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # (a 20-step run read 3 % faster than the 400-step soak median of its box)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="batch per GPU (BASELINE configs[1]: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--optimizer", default="flat", choices=["flat", "torch"],
                    help="flat: s2vt_adam_step over flat buffers (optim.FlatAdam, what train.py uses); torch: torch.optim.Adam(fused=True)")
    ap.add_argument("--decode-batch", type=int, default=None, help="greedy-decode batch (default 128 = BASELINE configs[4])")
    ap.add_argument("--gemm-mode", type=int, default=None, choices=[0, 1, 3],
                    help="0 fp32-input MFMA, 3 split-precision bf16x3 (default, fp32-equivalent), 1 bf16 operands "
                         "(BASELINE configs[2]: use with --batch 256)")
    ap.add_argument("--no-config3", action="store_true", help="skip the B=256 bf16 sub-record (BASELINE configs[2])")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed steps and the live kernel timing of the headline workload (no B=128 shard, isolated "
                         "pass, decode, beam, config3, CPU baseline): what tools/profile_round2.sh runs under rocprofv3, so "
                         "that the kernel-stats averages are those of the headline workload alone")
    ap.add_argument("--graphs", type=int, default=None, choices=[0, 1],
                    help="s2vt_set_graph_mode: replay the forward / backward launch sequences as hipGraphs (default: the library's, off)")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU-only rehearsal of the N-rank launch path (gloo); prints n_gpus")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher around us: start the N ranks ourselves (before anything touches the GPU) and exit with their code
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.selftest_launcher:
        return launcher_selftest(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_pg = world > 1 or ("RANK" in os.environ and os.environ.get("S2VT_BENCH_PG", "0") == "1")
    cu_reserved = 0
    if use_pg:      # one process per GPU over RCCL ("nccl" backend); a 1-rank group exercises the same code path
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import s2vt_video_caption_amd  # noqa: F401
        from s2vt_video_caption_amd import dp as _dp0
        cu_reserved = _dp0.plan_for_collectives(world)          # (before the communicator exists: NCCL_MAX_NCHANNELS + cu_reserve)
        dist.init_process_group(backend="nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("--gpus %d but the RCCL group has %d ranks" % (args.gpus, dist.get_world_size()))

    import S2VTModel
    import utils
    from s2vt_video_caption_amd import capi, dp, synth
    lib = capi.load()
    if args.gemm_mode is not None:
        lib.s2vt_set_gemm_mode(args.gemm_mode)
    mode = lib.s2vt_set_gemm_mode(-1)
    if args.graphs is not None:
        lib.s2vt_set_graph_mode(args.graphs)

    L, F, H, E, V = 80, 4096, 1000, 1000, 12000
    B = args.batch
    sd = synth.make_state_dict(V, F, H, E, seed=0)
    model = S2VTModel.S2VT(V, F, L, dim_hid=H, dim_embed=E)
    model.load_state_dict(sd)
    model.to(dev)
    crit = utils.MaskCriterion()
    reducer = dp.FlatGradAllReducer(model.parameters()).attach(model) if use_pg else None
    if args.optimizer == "flat":        # train.py:89-93's Adam as one launch over flat buffers (optim.py), what train.py of this repo uses
        from s2vt_video_caption_amd.optim import FlatAdam
        opt = FlatAdam(model, lr=1e-4, reducer=reducer)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)

    # this rank's shard of the synthetic global batch (seeded recipe); resident in HBM
    feats, caps, mask = synth.make_batch(B, L, F, V, seed=1234 + rank)
    feats, caps, mask = feats.to(dev), caps.to(dev), mask.to(dev)

    def one_step():
        return dp.train_step(model, crit, opt, feats, caps, mask, reducer)

    def sync_all():
        torch.cuda.synchronize(dev)
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize(dev)

    log("model and data resident; warm-up x%d" % args.warmup)
    for _ in range(args.warmup):
        one_step()
        torch.cuda.synchronize(dev)
        log("warm-up step done")
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    sync_all()
    dt = time.perf_counter() - t0
    dt_local = dt
    if use_pg:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    frames_per_s = world * B * L * args.steps / dt
    capi.check_async_error()      # a device-side error inside the timed steps (bad token, timed-out hand-off) voids the number
    final_loss = float(loss)
    # per-rank step time (which rank is the straggler?) and the gradient all-reduce under the microscope: three extra steps with
    # events around every gradient group's collective on the communication stream (dp.FlatGradAllReducer.read_timing) - how long
    # each group took, when it started relative to the end of the backward, how much of it the step had to wait for
    rank_ms, comm = None, None
    if use_pg:
        mine = torch.tensor([dt_local / args.steps * 1e3], dtype=torch.float64, device=dev)
        allms = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allms, mine)
        allms = [float(x.item()) for x in allms]
        rank_ms = {"min": round(min(allms), 3), "max": round(max(allms), 3), "by_rank": [round(x, 3) for x in allms]}
        reducer.timing = True
        for _ in range(3):
            one_step()
        sync_all()
        reducer.timing = False
        mine_comm = reducer.read_timing()
        gathered = [None] * world
        dist.all_gather_object(gathered, mine_comm)
        # (a rank that took the bucketed path reports {"path", "host_ms", ...}, one whose all_reduce returned early reports None:
        # the record is emitted as it stands, with its path label - nothing here may take the line down after the timed region)
        comm = dict(gathered[0]) if isinstance(gathered[0], dict) else {"path": "no timing record on rank 0"}
        comm["exposed_after_backward_ms_by_rank"] = [g.get("exposed_after_backward_ms") if isinstance(g, dict) else None for g in gathered]
        comm["group_ms_by_rank"] = [g.get("group_ms") if isinstance(g, dict) else None for g in gathered]
        comm["path_by_rank"] = [g.get("path") if isinstance(g, dict) else None for g in gathered]
        comm["note"] = ("events on the communication stream, mean of 3 steps after the timed region; group 0 out_linear, 1 word_rnn + "
                        "embedding, 2 vid_rnn + feat_linear; a group's start < 0 = issued while the backward was still running")
    # Host-side cost of ENQUEUING one step: the phases of a step timed separately with the queues drained before each (a
    # free-running host is throttled by queue back-pressure, which is GPU time, not host cost: round 2 reported that
    # figure, 8.3 ms, as if it were enqueue cost).  Must stay well below ms_per_step or the run is launch-bound.
    def host_cost(nrep=3):
        acc = {"zero_grad": 0.0, "forward": 0.0, "criterion": 0.0, "backward": 0.0, "allreduce+adam": 0.0}

        def ph(name, fn):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            r = fn()
            acc[name] += time.perf_counter() - t0
            return r
        for _ in range(nrep):
            ph("zero_grad", lambda: reducer.zero_grad() if reducer is not None else opt.zero_grad())
            model.train()
            probs = ph("forward", lambda: model(feats, targets=caps[:, :-1], mode="train"))
            ls = ph("criterion", lambda: crit(probs, caps, mask))
            ph("backward", lambda: ls.backward())
            ph("allreduce+adam", lambda: (reducer.all_reduce() if reducer is not None else None, opt.step()))
        torch.cuda.synchronize(dev)
        return {k: round(v / nrep * 1e3, 3) for k, v in acc.items()}
    host_phases = host_cost()
    host_ms = sum(host_phases.values())
    capi.check_async_error()
    log("timed region: %.3f ms/step, %.0f frames/s" % (ms_per_step, frames_per_s))

    # ---- BASELINE configs[3] shard: B=128 per GPU (global 1024 at 8 GPUs), same step, every rank takes part (sub-record;
    # the headline stays B per GPU so that N=1 agrees with the single-GPU line)
    shard128 = None
    if B != 128 and mode != 1 and not args.headline_only:
        f2, c2, m2 = synth.make_batch(128, L, F, V, seed=4321 + rank)
        f2, c2, m2 = f2.to(dev), c2.to(dev), m2.to(dev)
        for _ in range(2):
            dp.train_step(model, crit, opt, f2, c2, m2, reducer)
        sync_all()
        t0 = time.perf_counter()
        n128 = 50
        for _ in range(n128):
            dp.train_step(model, crit, opt, f2, c2, m2, reducer)
        sync_all()
        d128 = time.perf_counter() - t0
        capi.check_async_error()
        if use_pg:
            t = torch.tensor([d128], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d128 = float(t.item())
        shard128 = {"workload": "BASELINE configs[3] shard: B=128 per GPU x %d GPU = global %d" % (world, 128 * world),
                    "value": round(world * 128 * L * n128 / d128, 1), "unit": "frames/s", "ms_per_step": round(d128 / n128 * 1e3, 3),
                    "steps": n128, "global_batch": 128 * world}
        del f2, c2, m2
        log("B=128 shard: %s" % shard128)

    # ---- fixed GLOBAL batch (strong scaling; SURVEY.md 8(e) asks for it beside the weak-scaling headline): BASELINE configs[3]'s
    # global 1024 cut into 1024 / N rows per GPU - 128 per GPU at N = 8 - the same arithmetic, every rank taking part
    fixed_global = None
    if mode != 1 and not args.headline_only and 1024 % world == 0:
        perb, lo_, hi_ = dp.fixed_global_shard(1024, rank, world)
        fb = tuple(t[lo_:hi_].to(dev) for t in synth.make_batch(1024, L, F, V, seed=4321))
        nfg = 4 if perb >= 512 else 10
        for _ in range(2):
            dp.train_step(model, crit, opt, fb[0], fb[1], fb[2], reducer)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(nfg):
            dp.train_step(model, crit, opt, fb[0], fb[1], fb[2], reducer)
        sync_all()
        dfg = time.perf_counter() - t0
        capi.check_async_error()
        if use_pg:
            t = torch.tensor([dfg], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dfg = float(t.item())
        fixed_global = {"workload": "BASELINE configs[3] as strong scaling: global B = 1024 over %d GPU = %d per GPU" % (world, perb),
                        "scaling": "strong", "global_batch": 1024, "per_gpu_batch": perb, "value": round(1024 * L * nfg / dfg, 1),
                        "unit": "frames/s", "ms_per_step": round(dfg / nfg * 1e3, 3), "steps": nfg}
        del fb
        log("fixed global batch: %s" % fixed_global)

    out = None
    if rank == 0:
        # ---- live per-kernel timing with HIP events on the launch stream (extra steps, not part of `value`).
        # Pass 1 in the timed configuration (layers pipelined on two streams: kernels of the two lanes overlap,
        # so a launch's duration includes the share of the chip it cedes to the other lane); pass 2 with the
        # pipeline off (s2vt_set_pipeline_block(0)): every kernel alone on the GPU.
        def profile(nprof=2, batch=None):
            pf, pc, pm = batch if batch is not None else (feats, caps, mask)
            capi.check(lib.s2vt_prof_reset(), "prof_reset")
            capi.check(lib.s2vt_prof_enable(1), "prof_enable")
            for _ in range(nprof):
                model.zero_grad(set_to_none=False)
                probs = model(pf, targets=pc[:, :-1], mode="train")
                l2 = crit(probs, pc, pm)
                l2.backward()
            torch.cuda.synchronize(dev)
            capi.check_async_error()
            capi.check(lib.s2vt_prof_enable(0), "prof_enable")
            # (sum of bracket durations, launches, BUSY time = union of the brackets over both lanes): brackets of one
            # kind overlap where the two lanes run the same kind of kernel side by side (the weight-gradient GEMMs of the
            # two layers, the two layers' timesteps), and each overlapped launch lasts about twice as long as alone - the
            # sum counts that time twice, throughput is priced with the busy time
            r = {k: capi.prof_read(i) + (capi.prof_read_busy(i),) for i, k in ((0, "gemm"), (1, "step_fwd"), (2, "step_bwd"), (3, "ce"), (5, "gemm_corun"))}
            capi.check(lib.s2vt_prof_reset(), "prof_reset")
            return {k: (ms / nprof, n // nprof, busy / nprof) for k, (ms, n, busy) in r.items()}

        T = 2 * L - 1
        x3 = (mode == 3) and (B % 64 == 0)
        bf = (mode == 1) and (B % 64 == 0)
        esz = 2 if bf else 4
        gflop = gemm_flops_train(B, L, F, H, E, V) / 1e9
        pair_bytes = step_bytes_fwd(B, H, H, s=esz) + step_bytes_fwd(B, H, E + H, s=esz)

        # HBM traffic per launch from rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, tools/pmc_traffic.sh), committed
        # under profiles/ — counters cannot be collected from inside this process.  Only valid for the default workload.
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import source_digest

        def load_static(pattern, tag):
            """(kernels, provenance, all-kernel totals) of the newest committed counter file of this workload.  STATIC data (counters
            cannot be collected from inside this process): the file names the commit it was measured at and carries the sha1 of every
            kernel source of that tree; a kernel whose source changed since then is DROPPED (tools/source_digest.py), and a file
            without digests (collected before round 5) is quoted with `unverified: true`."""
            for rnd in ("round5", "round4"):
                path = os.path.join(ROOT, "profiles", pattern % (rnd, tag))
                try:
                    j = json.load(open(path))
                except Exception:
                    continue
                src = {"static": "profiles/%s@%s" % (os.path.basename(path), j.get("commit", "unrecorded"))}
                dig = j.get("source_digests")
                if not dig:
                    src["unverified"] = True
                    return j["kernels"], src, j.get("all_kernels")
                stale = {k: source_digest.stale_sources(dig, k) for k in j["kernels"]}
                dropped = sorted(k for k, v in stale.items() if v)
                if dropped:
                    src["dropped_as_stale"] = {k: stale[k] for k in dropped}
                tot = j.get("all_kernels") if not any(source_digest.digests().get(f) != h for f, h in dig.items()) else None
                return {k: v for k, v in j["kernels"].items() if not stale[k]}, src, tot
            return {}, None, None

        def load_pmc(tag):
            k, src, _ = load_static("%s_traffic_pmc_%s.json", tag)
            return k, src

        def load_busy(tag):
            """Matrix-pipe busy share and wave-state shares per kernel from the SQ counter passes (tools/profile_round5.sh pmc)"""
            k, src, _ = load_static("%s_pmc_%s.json", tag)
            return k, src

        def busy_of(table, src, kernel):
            """{mfma_busy (dispatch-weighted over the template instantiations), per-instantiation figures, provenance} for `kernel`."""
            inst = {k: v for k, v in table.items() if k.split("<")[0] == kernel and "mfma_busy" in v}
            if not inst:
                return None
            n = sum(v["dispatches"] for v in inst.values())
            return {"mfma_busy": round(sum(v["mfma_busy"] * v["dispatches"] for v in inst.values()) / max(n, 1), 4),
                    "by_instantiation": {k: {x: v.get(x) for x in ("mfma_busy", "wait_share", "issue_stall", "lds_conflict", "dispatches")}
                                         for k, v in inst.items()},
                    "source": src, "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) per dispatch"}
        pmc, pmc_src = ({}, None)
        busy2, busy2_src = load_busy("c2")
        busy3, busy3_src = load_busy("c3")
        busyd, busyd_src = load_busy("dec")
        if B == 64 and not bf:
            pmc, pmc_src = load_pmc("c2")
        elif B == 256 and bf:
            pmc, pmc_src = load_pmc("c3")

        plan = capi.recurrence_plan(B, H)          # (forward, BPTT) recurrence kernels of the timed configuration (s2vt_hip.h)
        persist_bf16 = plan[0] == 1
        FWD_KERNELS = {0: "lstm_step_fwd_kernel", 1: "lstm_seq_fwd_bf16_persist_kernel", 3: "lstm_seq_fwd_x3_persist_kernel"}
        BWD_KERNELS = {0: "lstm_step_bwd_kernel", 1: "lstm_seq_bwd_bf16_persist_kernel", 3: "lstm_seq_bwd_x3_persist_kernel"}

        def rooflines(pr, how, persist, B_=B, esz_=esz, bf_=bf, x3_=x3, pmc_=None, pmc_src_=None, busy_=None):
            # persist: (forward kind, BPTT kind) as s2vt_recurrence_plan reports them, or False for launches per timestep
            pf, pb = persist if isinstance(persist, tuple) else ((1, 1) if (persist and bf_) else (0, 0))
            pmc_ = pmc if pmc_ is None else pmc_
            pmc_src_ = pmc_src if pmc_src_ is None else pmc_src_

            def traffic_of(kernel):
                e = pmc_.get(kernel)
                if not e:
                    return None
                if "hbm_bytes_per_layer_timestep" in e:
                    return int(e["hbm_bytes_per_layer_timestep"])
                return int(e["hbm_bytes_per_launch"]) if "hbm_bytes_per_launch" in e else None
            gflop_ = gemm_flops_train(B_, L, F, H, E, V) / 1e9
            pair_ = step_bytes_fwd(B_, H, H, s=esz_) + step_bytes_fwd(B_, H, E + H, s=esz_)
            gemm_ms, gemm_n, gemm_busy = pr["gemm"]
            co_ms, co_n, co_busy = pr.get("gemm_corun", (0.0, 0, 0.0))
            # GEMMs of option corun run on PART of the compute units beside a one-layer persistent launch (which holds the rest): their
            # busy time is priced with the share of the device they were planned for; they never overlap a full-device GEMM
            co_share = 0.0
            if co_n:
                ncu = torch.cuda.get_device_properties(dev).multi_processor_count
                cap = (ncu - 128) // 8 * 8          # 126 persistent workgroups + 2 idle ones of its grid hold the rest
                co_share = max(min(cap / float(ncu), 1.0), 0.0)
            gemm_busy_w = gemm_busy + co_share * co_busy
            gemm_busy_all = gemm_busy + co_busy
            gemm_ms, gemm_n = gemm_ms + co_ms, gemm_n + co_n
            sf_ms, sb_ms = pr["step_fwd"][0], pr["step_bwd"][0]
            gemm_tf = gflop_ / gemm_busy_all          # GFLOP / ms = TFLOP/s over the time at least one GEMM was running
            step_us = sf_ms * 1e3 / (2 * T)           # average timestep of one layer, forward (2T per step: both layers)
            bstep_us = sb_ms * 1e3 / (2 * T)
            step_gbs = (pair_ / 2) / (step_us * 1e-6) / 1e9
            bstep_gbs = (pair_ / 2) / (bstep_us * 1e-6) / 1e9
            if bf_:
                gk, gpeak, gnote = "gemm_b1_kernel", MFMA_BF16_PEAK_TF, "bf16 operands, fp32 accumulate"
            elif x3_:   # each algorithmic fp32 product costs six bf16 MFMA products (three planes per operand)
                gk, gpeak = "gemm_x3_kernel", MFMA_BF16_PEAK_TF / 6.0
                gnote = ("achieved = algorithmic (fp32-equivalent) FLOP/s; peak = bf16 dense MFMA peak / 6 plane products; "
                         "executed MFMA rate = 6 x achieved")
            else:
                gk, gpeak, gnote = "gemm_f32_kernel", MFMA_F32_PEAK_TF, "fp32-input MFMA"
            gnote += ("; achieved = algorithmic FLOPs of all GEMM launches of a step / BUSY time (union of the launch brackets: "
                      "GEMMs of the two lanes that run side by side are counted once); `sum_of_launch_ms` counts overlapped "
                      "launches twice and is what a per-kernel profile (rocprofv3 --stats) adds up to")
            if co_n:
                gnote += ("; option corun: %d of the %d launches run on %.0f %% of the compute units beside the one-layer launches of the "
                          "persistent recurrence (which hold the rest): `achieved` / `frac` price their busy time (%.3f ms per step) in full, "
                          "as a per-kernel profile does; `frac_by_device_share` prices it with the share of the device those launches were "
                          "planned for - the figure that compares with earlier rounds' full-device GEMMs (profiles/round5_corun.txt)"
                          % (co_n, gemm_n, 100 * co_share, co_busy))
            rg = {"kernel": gk, "bound": "mfma", "achieved": round(gemm_tf, 2),
                  "peak": round(gpeak, 1), "unit": "TFLOP/s", "frac": round(gemm_tf / gpeak, 4),
                  "traffic": traffic_of(gk), "traffic_source": pmc_src_, "launches_per_step": gemm_n,
                  "busy_ms_per_step": round(gemm_busy_all, 3), "busy_ms_full_device": round(gemm_busy, 3),
                  "busy_ms_part_of_device": round(co_busy, 3), "device_share_of_those": round(co_share, 3),
                  "frac_by_device_share": round(gflop_ / gemm_busy_w / gpeak, 4), "sum_of_launch_ms": round(gemm_ms, 3),
                  "frac_by_sum_of_launch_ms": round(gflop_ / gemm_ms / gpeak, 4),
                  "algorithmic_bytes_or_flops_per_launch": round(gflop_ * 1e9 / max(gemm_n, 1)),
                  "algorithmic_gflop_per_step": round(gflop_, 1), "timing": how, "note": gnote,
                  "pipe_counters": busy_of(*busy_, gk) if busy_ else None}
            if pf:
                fk = FWD_KERNELS[pf]
                fnote = ("PERSISTENT-WEIGHTS kernel: one launch runs a block of timesteps of BOTH layers with every W_hh slice "
                         "resident in registers, so W is not re-streamed and the fraction may exceed 1 (SURVEY.md §8(d)); "
                         "achieved = §8(d) ALGORITHMIC bytes of a (vid, word) timestep pair / 2 over the average per-layer timestep "
                         "= launch duration / timesteps in the launch: an equivalent-streaming rate, not bytes that moved - "
                         "`hbm_gbs_measured` is PMC traffic / the same time")
            else:
                fk = "lstm_step_fwd_bf16_kernel" if bf_ else "lstm_step_fwd_kernel"
                fnote = ("bytes per SURVEY.md §8(d) incl. W_ih although the x-part is hoisted into a batched GEMM; avg over vid+word "
                         "launches, loop-bracketed events (includes launch gaps)")
            if pb:
                bk = BWD_KERNELS[pb]
            else:
                bk = "lstm_step_bwd_bf16_kernel" if bf_ else "lstm_step_bwd_kernel"

            def step_rec(kernel, gbs, us, note, busy_ms):
                tr = traffic_of(kernel)
                busy_us = busy_ms * 1e3 / (2 * T)          # wall time per layer timestep while at least one lane runs this family
                return {"kernel": kernel, "bound": "hbm", "achieved": round(gbs, 1),
                        "busy_us_per_layer_timestep": round(busy_us, 3),
                        "frac_by_busy_time": round((pair_ / 2) / (busy_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "traffic": tr, "traffic_source": pmc_src_ if tr is not None else None,
                        "hbm_gbs_measured": round(tr / (us * 1e-6) / 1e9, 1) if tr is not None else None,
                        "avg_launch_us": round(us, 3), "launches_per_step": 2 * T,
                        "ms_per_step": round(us * 2 * T / 1e3, 3),
                        "algorithmic_bytes_per_launch": pair_ // 2, "timing": how, "note": note,
                        "pipe_counters": busy_of(*busy_, kernel) if busy_ else None}
            lane_note = ("; `frac` prices the average LAUNCH (two lanes' launches overlap and stretch each other), `frac_by_busy_time` the "
                         "wall time per layer timestep while the family runs (union of the lanes' brackets)")
            rs = step_rec(fk, step_gbs, step_us, fnote + lane_note, pr["step_fwd"][2])
            rb = step_rec(bk, bstep_gbs, bstep_us,
                          "BPTT timestep: dh = dG_{t+1} W_hh + gate derivatives; same §8(d) byte accounting as the forward timestep" + lane_note +
                          ("; option bptt_solo: every launch carries ONE layer on half of the compute units (the layers run one after the other: "
                           "twice the launches of the two-layer schedule, 0.56 there) while the backward's GEMMs fill the other half - the step is "
                           "0.4 ms shorter, this family's own time longer (profiles/round5_corun.txt)"
                           if (co_n and pb == 3 and int(lib.s2vt_set_option(b"bptt_solo", -1))) else ""),
                          pr["step_bwd"][2])
            return rg, rs, rb

        live = profile()
        prev_blk = lib.s2vt_set_pipeline_block(0)          # (returns the block in use ...
        lib.s2vt_set_pipeline_block(prev_blk)              # ... and this puts it back)
        eff_blk = prev_blk           # api.hip balanced_block(): the persistent bf16 schedule evens the default 32 out over the L frames
        if plan[0] in (1, 3) and prev_blk == 32:
            eff_blk = -(-L // -(-L // 32))
        busy_here = (busy3, busy3_src) if (B == 256 and bf) else (busy2, busy2_src) if (B == 64 and not bf) else None
        roof_gemm, roof_step, roof_bstep = rooflines(live, "live, layers pipelined (block %d)" % eff_blk, plan, busy_=busy_here)
        if args.headline_only:
            alone = live
            roof_gemm_alone = roof_step_alone = roof_bstep_alone = None
        else:
            lib.s2vt_set_pipeline_block(0)
            alone = profile()
            lib.s2vt_set_pipeline_block(prev_blk)
            # pipeline off = one launch per timestep, every kernel alone on the GPU (the persistent kernels need the block schedule)
            roof_gemm_alone, roof_step_alone, roof_bstep_alone = rooflines(alone, "pipeline off: launch per timestep, every kernel alone on the GPU", False)
        log("profiled steps done (pipeline block %d)" % prev_blk)
        gname = roof_gemm["kernel"]
        gsum = lambda pr_: pr_["gemm"][0] + pr_.get("gemm_corun", (0.0,))[0]      # all GEMM launches, full-device and co-run ones
        fam = {gname: gsum(live), roof_step["kernel"]: live["step_fwd"][0],
               roof_bstep["kernel"]: live["step_bwd"][0], "ce": live["ce"][0]}
        fam_alone = None if args.headline_only else {gname: gsum(alone), roof_step_alone["kernel"]: alone["step_fwd"][0],
                                                     roof_bstep_alone["kernel"]: alone["step_bwd"][0], "ce": alone["ce"][0]}
        # The headline `roofline` is the kernel family with the LARGEST live time in THIS run (sum of its launch durations, what
        # a per-kernel profile ranks by); the other two families follow as roofline_gemm / roofline_lstm_step[_bwd], and
        # `roofline_min` names the weakest fraction of the three.
        cands = [roof_gemm, roof_step, roof_bstep]
        live_ms = [gsum(live), live["step_fwd"][0], live["step_bwd"][0]]
        roofline = dict(cands[max(range(3), key=lambda i: live_ms[i])])
        roofline["chosen_as"] = "largest sum of launch durations in this run"
        roofline["family_ms_per_step"] = {"batched_gemm": round(live_ms[0], 3), "timestep_fwd": round(live_ms[1], 3),
                                          "timestep_bwd": round(live_ms[2], 3)}
        weakest = min(cands, key=lambda r: r["frac"])
        roofline_min = {"kernel": weakest["kernel"], "frac": weakest["frac"], "bound": weakest["bound"]}

        decode = beam = None
        if not args.headline_only:
            # ---- greedy decode captions/s (one mode='test' call per measurement) + its out_linear/argmax kernel
            Bd = args.decode_batch or 128        # BASELINE configs[4]: inference at B=128
            dfe = feats[:Bd] if Bd <= B else synth.make_batch(Bd, L, F, V, seed=99)[0].to(dev)
            model.load_state_dict(sd)            # the seeded weights again (the train steps above moved them): the decode and beam
            model.eval()                         # legs are then the same computation in every run
            from s2vt_video_caption_amd import functional as _fn
            with torch.no_grad():
                model(dfe, mode="test")
                torch.cuda.synchronize(dev)
                # cold: every call rebuilds the weight-derived images (first call after the weights changed)
                _fn.clear_decode_cache(model)
                keep_cache = _fn.DECODE_CACHE
                _fn.DECODE_CACHE = False
                t1 = time.perf_counter()
                for _ in range(3):
                    model(dfe, mode="test")
                torch.cuda.synchronize(dev)
                cold_dt = (time.perf_counter() - t1) / 3
                _fn.DECODE_CACHE = keep_cache
                model(dfe, mode="test")                         # fills the cache
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                nd = 10
                for _ in range(nd):
                    ids = model(dfe, mode="test")               # eval.py's regime: fixed weights, one call per batch
                torch.cuda.synchronize(dev)
                ddt = (time.perf_counter() - t1) / nd
                capi.check(lib.s2vt_prof_reset(), "prof_reset")
                capi.check(lib.s2vt_prof_enable(1), "prof_enable")
                model(dfe, mode="test")
                torch.cuda.synchronize(dev)
                capi.check(lib.s2vt_prof_enable(0), "prof_enable")
                am_ms, am_n = capi.prof_read(4)
                capi.check(lib.s2vt_prof_reset(), "prof_reset")
            am_us = am_ms * 1e3 / max(am_n, 1)
            am_bytes = 4 * V * H + 4 * V + 4 * Bd * H + 8 * Bd         # W_o + b_o + h + packed argmax words
            am_gflop = 2.0 * Bd * V * H / 1e9
            planes = (mode != 0) and Bd % 64 == 0                      # the plane-path kernel (csrc/argmax_x3.hip)
            fused = planes and lib.s2vt_set_decode_schedule(-1) == 1 and am_n == L
            if fused:
                # fused schedule: L launches do the L-1 argmax steps AND the L-1 recurrent GEMMs h_t W_hh^T (4H more rows of the
                # same kernel; the first launch has only those, the last only the vocabulary's): FLOPs and bytes averaged per launch
                am_gflop = 2.0 * Bd * (V + 4 * H) * H * (L - 1) / L / 1e9
                am_bytes = int((4 * (V + 4 * H) * H + 4 * V + 4 * Bd * H + 8 * Bd + 16 * Bd * H) * (L - 1) / L)
            am_peak = (MFMA_BF16_PEAK_TF / 6.0) if planes else MFMA_F32_PEAK_TF
            am_kernel = "logits_argmax_x3_kernel" if planes else "logits_argmax_kernel"
            pmcd, pmcd_src = load_pmc("dec")             # one cold greedy decode at B = 128 (tools/profile_round5.sh traffic)
            decode = {"metric": "greedy-decode captions/sec", "value": round(Bd / ddt, 1), "unit": "captions/s",
                      "batch": Bd, "ms_per_call": round(ddt * 1e3, 2), "n_gpus": 1, "calls_timed": nd,
                      "regime": "fixed weights, one mode='test' call per batch (eval.py:48-52): weight-derived images cached between calls",
                      "cold_ms_per_call": round(cold_dt * 1e3, 2), "cold_captions_per_s": round(Bd / cold_dt, 1),
                      "roofline_logits_argmax": {
                          "kernel": "logits_argmax_x3_kernel" if planes else "logits_argmax_kernel", "bound": "mfma",
                          "achieved": round(am_gflop / (am_us * 1e-6) / 1e3, 1), "peak": round(am_peak, 1), "unit": "TFLOP/s",
                          "frac": round(am_gflop / (am_us * 1e-6) / 1e3 / am_peak, 4),
                          "avg_launch_us": round(am_us, 2), "launches_per_call": am_n,
                          "gflop_per_launch": round(am_gflop, 2),
                          "traffic": (int(pmcd[am_kernel]["hbm_bytes_per_launch"]) if Bd == 128 and am_kernel in pmcd and
                                      "hbm_bytes_per_launch" in pmcd[am_kernel] else None),
                          "traffic_source": pmcd_src if Bd == 128 else None,
                          "pipe_counters": busy_of(busyd, busyd_src, "logits_argmax_x3_kernel" if planes else "logits_argmax_kernel") if Bd == 128 else None,
                          "algorithmic_bytes_per_launch": am_bytes,
                          "hbm_frac_by_algorithmic_bytes": round(am_bytes / (am_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                          "schedule": "fused (argmax of step t + h_t W_hh^T of step t+1 in one launch)" if fused else "step kernel + argmax kernel",
                          "note": ("fp32-equivalent FLOP/s of h [W_o; W_hh]^T against the bf16 dense MFMA peak / 6 plane products (3 bf16 planes "
                                   "per operand); the weight planes are written once per weight version, h_t planes by the cell-update kernel"
                                   if planes else "fp32-input MFMA kernel (batches that are not multiples of 64)")}}
            # ---- beam search (BASELINE configs[4]: B=128, beam_size 5, depth 30)
            # The interpreter's cyclic collector is run before the leg: a generation-2 pass over what the training leg left
            # (~75 ms, profiles/round4_beam_first_call_after_training.txt) otherwise lands inside the first timed call, whose 3000
            # result tensors trip the allocation counter.  Every call is timed on its own; value is the MEAN of all of them.
            import gc
            with torch.no_grad():
                model(dfe, mode="beam_search", beam_width=5, max_beam_depth=30)
                torch.cuda.synchronize(dev)
                gc.collect()
                nbm = 10
                calls = []
                for _ in range(nbm):
                    t1 = time.perf_counter()
                    model(dfe, mode="beam_search", beam_width=5, max_beam_depth=30)
                    torch.cuda.synchronize(dev)
                    calls.append(time.perf_counter() - t1)
                bdt = sum(calls) / nbm
            from s2vt_video_caption_amd import beam as _beam
            beam = {"metric": "beam-search captions/sec (beam 5, depth 30)", "value": round(Bd / bdt, 1), "unit": "captions/s",
                    "batch": Bd, "ms_per_call": round(bdt * 1e3, 2), "n_gpus": 1, "calls_timed": nbm, "path": _beam.LAST_PATH,
                    "ms_per_call_min_max": [round(min(calls) * 1e3, 2), round(max(calls) * 1e3, 2)]}
            model.train()
            capi.check_async_error()

        # ---- BASELINE configs[2] in the same run (N = 1): B=256, bf16 operands / fp32 accumulate (s2vt_set_gemm_mode(1)),
        # persistent recurrence kernels - the configuration north_star puts its roofline target on
        config3 = None
        if world == 1 and not bf and B != 256 and not args.no_config3 and not args.headline_only:
            prev_mode = lib.s2vt_set_gemm_mode(1)
            try:
                b3 = tuple(t.to(dev) for t in synth.make_batch(256, L, F, V, seed=777))
                for _ in range(2):
                    dp.train_step(model, crit, opt, b3[0], b3[1], b3[2], None)
                torch.cuda.synchronize(dev)
                t3 = time.perf_counter()
                n3 = 50
                for _ in range(n3):
                    dp.train_step(model, crit, opt, b3[0], b3[1], b3[2], None)
                torch.cuda.synchronize(dev)
                d3 = (time.perf_counter() - t3) / n3
                capi.check_async_error()
                pr3 = profile(2, b3)
                pers = lib.s2vt_set_recurrence_mode(-1) >= 1
                pmc3, pmc3_src = load_pmc("c3")
                g3, f3, bw3 = rooflines(pr3, "live, persistent recurrence (block %d)" % (-(-L // -(-L // 32)) if prev_blk == 32 else prev_blk) if pers else "live", pers,
                                        B_=256, esz_=2, bf_=True, x3_=False, pmc_=pmc3, pmc_src_=pmc3_src, busy_=(busy3, busy3_src))
                config3 = {"workload": "BASELINE configs[2]: B=256, 80x4096 feats, hidden=embed=1000, vocab=12000, bf16 operands / "
                                       "fp32 accumulate, Adam", "dtype": "bf16", "value": round(256 * L / d3, 1), "unit": "frames/s",
                           "ms_per_step": round(d3 * 1e3, 3), "steps": n3,
                           "roofline_lstm_step": f3, "roofline_lstm_step_bwd": bw3, "roofline_gemm": g3,
                           "gemm_tflops": g3["achieved"],
                           "kernel_ms_per_step": {k: round(v[0], 3) for k, v in pr3.items()},
                           "kernel_busy_ms_per_step": {k: round(v[2], 3) for k, v in pr3.items()}}
                del b3
            finally:
                lib.s2vt_set_gemm_mode(prev_mode)
            log("config3: %s" % config3)
        # ---- the reference's OWN defaults (train.py:27-28,37: batch_size = 16, dim_hidden = dim_embed = 512; eval.py:27: batch 10):
        # ragged batches are padded to a multiple of 64 inside the library's workspace from 33 rows on (s2vt_padded_batch); these
        # sizes run as they are on the launch-per-timestep path, which measured faster - the forced-padding time and the
        # reference's CPU path stand beside the number
        ref_defaults = None
        if world == 1 and not bf and not args.headline_only:
            Hd, Bt, Be = 512, 16, 10
            sd_d = synth.make_state_dict(V, F, Hd, Hd, seed=5)
            md = S2VTModel.S2VT(V, F, L, dim_hid=Hd, dim_embed=Hd)
            md.load_state_dict(sd_d)
            md.to(dev)
            if args.optimizer == "flat":
                optd = FlatAdam(md, lr=1e-4)
            else:
                optd = torch.optim.Adam(md.parameters(), lr=1e-4, fused=True)
            bd_ = tuple(t.to(dev) for t in synth.make_batch(Bt, L, F, V, seed=778))
            fe = synth.make_batch(Be, L, F, V, seed=779)[0].to(dev)

            def timed(fn, n):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize(dev)
                t_ = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize(dev)
                capi.check_async_error()
                return (time.perf_counter() - t_) / n
            step_d = lambda: dp.train_step(md, crit, optd, bd_[0], bd_[1], bd_[2], None)

            def dec_d():
                with torch.no_grad():
                    md(fe, mode="test")
            t_plane = timed(step_d, 30)                      # the library's own choice for these sizes (option pad_min_batch)
            md.eval(); d_plane = timed(dec_d, 10); md.train()
            prev_pad = lib.s2vt_set_option(b"pad_min_batch", 1)      # forced: batch padded to 64, plane path
            try:
                t_f32 = timed(step_d, 10)
                md.eval(); d_f32 = timed(dec_d, 5); md.train()
            finally:
                lib.s2vt_set_option(b"pad_min_batch", prev_pad)
            cpu_d = None
            if not args.no_cpu_baseline:
                from oracle import s2vt_oracle as orc
                torch.set_num_threads(usable_cores())
                cmd_ = orc.ReferenceShapedCPUModel(sd_d)
                coptd = torch.optim.Adam(cmd_.parameters(), lr=1e-4)
                cfd, ccd, ckd = (t.cpu() for t in bd_)

                def cpu_step_d():
                    coptd.zero_grad()
                    orc.mask_criterion(cmd_(cfd, ccd[:, :-1]), ccd, ckd).backward()
                    coptd.step()
                cpu_step_d()
                t_ = time.perf_counter()
                for _ in range(3):
                    cpu_step_d()
                cdt_d = (time.perf_counter() - t_) / 3
                t_ = time.perf_counter()
                cmd_.greedy(fe.cpu())
                gdt_d = time.perf_counter() - t_
                cpu_d = {"value": round(Bt * L / cdt_d, 1), "unit": "frames/s", "cores": usable_cores(), "kind": "port",
                         "ms_per_step": round(cdt_d * 1e3, 1), "greedy_captions_per_s": round(Be / gdt_d, 2),
                         "sample": "B=16 train steps (1 warm-up + 3 timed) and one B=10 greedy decode with torch-CPU nn.LSTM / nn.Linear"}
            ref_defaults = {"workload": "the reference's defaults: train B=16, hidden=embed=512 (train.py:27-28,37), greedy decode B=10 "
                                        "(eval.py:27); 80x4096 feats, vocab=12000, fp32-equivalent, Adam",
                            "runs_at_batch": int(lib.s2vt_padded_batch(Bt)),
                            "path": "launches per timestep at the batch as it is (fp32-MFMA tiles): faster than 64 padded rows on the plane "
                                    "path below 33 rows (train) / 24 clips (decode), profiles/round5_ragged_batches.txt",
                            "train": {"value": round(Bt * L / t_plane, 1), "unit": "frames/s", "ms_per_step": round(t_plane * 1e3, 3),
                                      "ms_per_step_padded_to_64_plane_path": round(t_f32 * 1e3, 3)},
                            "decode": {"value": round(Be / d_plane, 1), "unit": "captions/s", "ms_per_call": round(d_plane * 1e3, 3),
                                       "ms_per_call_padded_to_64_plane_path": round(d_f32 * 1e3, 3)},
                            "cpu_baseline": cpu_d}
            del md, optd, bd_, fe
            log("ref_defaults: %s" % ref_defaults)
        log("decode: %s" % decode)
        cpu = None
        if world == 1 and not args.no_cpu_baseline and not args.headline_only:
            from oracle import s2vt_oracle as orc
            cores = usable_cores()
            log("cpu_baseline on %d threads" % cores)
            torch.set_num_threads(cores)
            Bc = min(64, B)          # the full BASELINE configs[1] batch: ~2-3 s per step on 16 cores
            cm = orc.ReferenceShapedCPUModel(sd)
            copt = torch.optim.Adam(cm.parameters(), lr=1e-4)
            cf, cc, ck = feats[:Bc].cpu(), caps[:Bc].cpu(), mask[:Bc].cpu()

            def cpu_step():
                copt.zero_grad()
                lg = cm(cf, cc[:, :-1])
                ls = orc.mask_criterion(lg, cc, ck)
                ls.backward()
                copt.step()
            cpu_step()
            t2 = time.perf_counter()
            ncpu = 2
            for _ in range(ncpu):
                cpu_step()
            cdt = (time.perf_counter() - t2) / ncpu
            t3 = time.perf_counter()
            cm.greedy(cf)
            gdt = time.perf_counter() - t3
            cpu = {"value": round(Bc * L / cdt, 1), "unit": "frames/s", "cores": cores, "kind": "port",
                   "sample": "same dims (L=80,F=4096,H=E=1000,V=12000), B=%d rows of the same batch, 1 warm-up + %d "
                             "timed train steps with torch-CPU nn.LSTM/nn.Linear (what the reference runs on CPU), "
                             "Adam(lr=1e-4); greedy: one call" % (Bc, ncpu),
                   "ms_per_step": round(cdt * 1e3, 1), "greedy_captions_per_s": round(Bc / gdt, 2)}

        # HBM-side bytes of a whole optimisation step (every dispatch, PMC passes of tools/profile_round5.sh traffic) beside what the step
        # has to move whatever the kernels do: inputs, logits out and back in, parameters in, gradients out, Adam (28 B / parameter)
        hbm_step = None
        if world == 1 and ((B == 64 and x3) or (B == 256 and bf)):
            _, tsrc, tot = load_static("%s_traffic_pmc_%s.json", "c2" if x3 else "c3")
            n_par = sum(p_.numel() for p_ in model.parameters())
            compulsory = 4 * (B * L * F + 2 * B * (L - 1) * V + 2 * n_par) + 28 * n_par
            hbm_step = {"pmc_bytes": int(tot["hbm_bytes_per_pass"]) if tot else None,
                        "pmc_gbs_over_the_step": round(tot["hbm_bytes_per_pass"] / (dt / args.steps) / 1e9, 1) if tot else None,
                        "source": tsrc if tot else ({"dropped": "a kernel source changed since the counters were collected", **(tsrc or {})}),
                        "compulsory_bytes": int(compulsory),
                        "algorithmic_bytes_8d_recurrence": int(2 * T * pair_bytes / 2),
                        "note": "pmc_bytes: FETCH_SIZE x 2 + WRITE_SIZE of every dispatch of one step; compulsory_bytes: feats read, logits "
                                "written and read, parameters read, gradients written, Adam's 28 B per parameter; algorithmic_bytes_8d_recurrence: "
                                "SURVEY 8(d)'s streaming model of the 318 layer timesteps, which the persistent kernels do not move (W_hh stays in registers)"}
        out = {
            "metric": "training frames/sec (whole node)", "value": round(frames_per_s, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf else "f32",
            "data": "synthetic",
            "arithmetic": ("fp32 storage and accumulation; batched GEMM products as 3 bf16 planes x 6 plane products on the bf16 "
                           "matrix cores (fp32-equivalent, ~2^-23 relative) - the batched GEMMs and every persistent recurrence kernel "
                           "of s2vt_recurrence_plan (kind 3); launch-per-timestep recurrences (kind 0) on the exact fp32-input MFMA"
                           if x3 else ("bf16 operands (weights, activations) on the bf16 matrix cores for batched and recurrent "
                                       "GEMMs; fp32 accumulation, cell state, gate stash, gradients, master weights and Adam"
                                       if bf else "fp32 storage, fp32-input MFMA, fp32 accumulation")),
            "config": {"workload": ("BASELINE configs[2]: S2VT train step, B=%d per GPU x %d GPU, 80x4096 feats, hidden=embed=1000, "
                                    "vocab=12000, bf16 operands / fp32 accumulate, Adam" % (B, world)) if bf else
                                   ("BASELINE configs[1]: S2VT train step, B=%d per GPU x %d GPU, 80x4096 feats, "
                                    "hidden=embed=1000, vocab=12000, fp32, Adam" % (B, world)),
                       "global_batch": B * world, "frames": L, "parallelism": "dp%d" % world},
            "final_loss": round(final_loss, 6), "host_enqueue_ms_per_step": round(host_ms, 3),
            "host_enqueue_ms_by_phase": host_phases, "graph_mode": int(lib.s2vt_set_graph_mode(-1)),
            "pipeline_streams_overlap": int(lib.s2vt_pipeline_overlaps()),
            "roofline": roofline,
            "roofline_min": roofline_min,
            "roofline_gemm": roof_gemm,
            "roofline_lstm_step": roof_step,
            "roofline_lstm_step_bwd": roof_bstep,
            "roofline_isolated": None if args.headline_only else {"gemm": roof_gemm_alone, "lstm_step": roof_step_alone, "lstm_step_bwd": roof_bstep_alone},
            "kernel_ms_per_step": {k: round(v, 3) for k, v in fam.items()},
            "kernel_busy_ms_per_step": {"gemm": round(live["gemm"][2], 3), "gemm_beside_one_layer_launches": round(live.get("gemm_corun", (0, 0, 0.0))[2], 3),
                                        "step_fwd": round(live["step_fwd"][2], 3),
                                        "step_bwd": round(live["step_bwd"][2], 3), "ce": round(live["ce"][2], 3)},
            "kernel_ms_per_step_isolated": {k: round(v, 3) for k, v in fam_alone.items()} if fam_alone else None,
            "hbm_bytes_per_step": hbm_step,
            "recurrence_plan": {"forward": plan[0], "bptt": plan[1], "kinds": "0 launch per timestep, 1 persistent bf16, 3 persistent split precision"},
            "optimizer": "optim.FlatAdam (s2vt_adam_step: torch.optim.Adam's arithmetic, one launch over flat buffers)" if args.optimizer == "flat"
                         else "torch.optim.Adam(fused=True)",
            "options": {lib.s2vt_option_name(i).decode(): int(lib.s2vt_set_option(lib.s2vt_option_name(i), -1)) for i in range(lib.s2vt_option_count())},
            "decode": decode,
            "beam": beam,
            "cpu_baseline": cpu,
            "ms_per_step_by_rank": rank_ms,
            "comm": comm,
            "rccl_ranks": dist.get_world_size() if use_pg else 1,
            "rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()) if use_pg else None,
            "dp_b128": shard128,
            "fixed_global_1024": fixed_global,
            "cu_reserved_for_collectives": cu_reserved,
            "config3": config3,
            "ref_defaults": ref_defaults,
        }
        print(json.dumps(out), flush=True)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
