#!/usr/bin/env python
"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

metric   : NMF MultUpdate-MSE outer iterations per second (and GFLOP/s) on X = 16384 x 16384, k = 256, fp32
step     : ONE outer iteration of update_wh!(::MultUpdMSE) (src/multupd.jl:83-116) + stop_condition statistics
           (src/common.jl:92-111), on synthetic dense X already resident in HBM.
N GPUs   : X and H are column-sharded over the ranks (same global X: strong scaling).  W side per iteration (default
           --comm-mode row_sharded): RCCL reduce-scatter of X_g H_g' by row blocks (+ all-reduce of the k x k / k-vector tail),
           each rank updates ITS rows of W, all-gather of W.  One process per GPU (torch.distributed.run).

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     : dominant kernel (the two p*n*k MFMA GEMMs), algorithmic flops per launch / hipEvent-measured
                 average launch duration inside the timed region, vs the 157.3 TFLOP/s fp32 MFMA peak.
  cpu_baseline : the NumPy restatement of the reference (oracle/nmf_oracle.py, "port", executes the reference's
                 6-GEMM sequence with multithreaded OpenBLAS like stock Julia) timed on the host cores on a bounded
                 column sample of the same workload; rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nmf.jl_amd"))

import numpy as np
import torch

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
PEAK_HBM_GBS = 8000.0
SEED = 20240910


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--p", type=int, default=16384)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--k", type=int, default=256)
    ap.add_argument("--alg", default="multmse", choices=["multmse", "multdiv", "projals", "alspgrad", "cd", "greedycd"])
    ap.add_argument("--maxsubiter", type=int, default=200, help="alspgrad: inner iteration cap per sub-solve (ALSPGrad.maxsubiter, reference default 200)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="bf16x3: OPT-IN mixed-precision form of the two p*n*k products (three bf16 MFMA products per term, fp32 "
                         "accumulation); not the headline configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not bracket launches with hipEvents (no roofline object)")
    ap.add_argument("--all-events", action="store_true", help="hipEvent pair around every launch (per-kernel table; slows the loop ~4%%)")
    ap.add_argument("--cpu-sample-cols", type=int, default=2048)
    ap.add_argument("--cpu-full-iters", type=int, default=2,
                    help="multmse: after the bounded column sample, also time this many iterations of the CPU port on the FULL problem (own process, "
                         "state pre-touched); 0 = scaled sample only")
    ap.add_argument("--check-every", type=int, default=1 << 30,
                    help="iterations between the host's polls of the device stop flag (default: never inside the timed region; the "
                         "API default is 0 = adaptive -- small problems pay a host round trip per poll)")
    ap.add_argument("--sim-ranks", type=int, default=0,
                    help="measurement aid (1 GPU): time rank 0's COMPUTE of an N-rank run -- X, H are the rank's column shard, the "
                         "collectives move their bytes device-locally (results are not a factorisation; never a headline number)")
    ap.add_argument("--comm-mode", default="row_sharded", choices=["row_sharded", "replicated_w", "pipelined"],
                    help="multi-GPU W side: reduce-scatter / row-sharded update / all-gather (default) or one packed all-reduce + replicated update")
    ap.add_argument("--transport", default="p2p", choices=["p2p", "rccl", "p2p_only"],
                    help="multi-GPU exchange: p2p (default) = the ranks' peer-mapped windows with device-side flags (all xGMI links at once), RCCL for "
                         "the bootstrap and as the fallback -- verified against an RCCL run of the same iterations before the timed region and "
                         "dropped for plain RCCL if it fails; rccl = RCCL collectives; p2p_only = windows without any RCCL communicator")
    ap.add_argument("--traffic", default="live", choices=["live", "static", "none"],
                    help="roofline.traffic of the default workload: live = two short rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of "
                         "this script, ~20 s each) behind the timed region; static = the value committed in profiles/pmc_traffic.json; falls back to "
                         "static when rocprofv3 is missing or a pass fails")
    ap.add_argument("--prewarm-ms", type=float, default=80.0,
                    help="device pre-warm BEFORE the W warm-up steps: untimed iterations worth about this many ms (the first ~50 ms after "
                         "idle run ~2%% slower while the clocks ramp; with W = 5 warm-up steps = 10 ms the timed region would sit on the ramp). "
                         "0 disables.  The factors are reset to the start values afterwards; the count is reported as `prewarm_steps`")
    ap.add_argument("--select", default="fastest", choices=["fastest", "requested"],
                    help="multi-GPU: fastest (default) = EVERY exchange candidate (the requested transport / mode, peer windows, RCCL row-sharded, RCCL "
                         "replicated-W, RCCL pipelined) is verified AND timed (--select-iters iterations after as many warm-up iterations, max over "
                         "ranks) before the timed region, which then runs on the fastest verified one; requested = the first verified candidate in "
                         "ladder order, starting with --transport / --comm-mode (round 5 behaviour)")
    ap.add_argument("--select-iters", type=int, default=20, help="iterations per candidate timing (and as many untimed ones in front)")
    ap.add_argument("--watchdog-s", type=float, default=600.0,
                    help="multi-GPU: a stage that takes longer than this prints a JSON line with status = comm_timeout and exits (a mismatched "
                         "collective would otherwise hang until the driver's timeout and leave no line at all)")
    # (ranks started by self_launch() find the original command line in NMFX_BENCH_ARGV)
    return ap.parse_args(json.loads(os.environ["NMFX_BENCH_ARGV"]) if ("NMFX_BENCH_ARGV" in os.environ and len(sys.argv) == 1) else None)


class Watchdog:
    """A stage of a multi-rank run that does not finish (a collective some rank never enters) must still produce a line."""

    def __init__(self, seconds, rank, base):
        self.seconds, self.rank, self.base, self.timer, self.stage = seconds, rank, base, None, None

    def arm(self, stage):
        import threading
        self.disarm()
        self.stage = stage
        self.timer = threading.Timer(self.seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None

    def _fire(self):
        if self.rank == 0:
            print(json.dumps(dict(self.base, value=None, status="comm_timeout", stage=self.stage, watchdog_s=self.seconds)), flush=True)
        os._exit(3)


def synth(p, n, k, c0, c1, tdtype, device, normalize_w0=True, warm=0.0):
    """Planted-rank dense X >= 0: X = Wg Hg + 0.01 U, generated on the device.
    Returns X^T shard as an (n_local, p) row-major tensor == column-major p x n_local, plus host W0, H0 shard."""
    g = torch.Generator(device="cpu")
    g.manual_seed(SEED)
    Wg = torch.rand((p, k), generator=g, dtype=torch.float32)
    Hg = torch.rand((k, n), generator=g, dtype=torch.float32)
    W0 = torch.rand((p, k), generator=g, dtype=torch.float64)
    if warm > 0:
        W0 = Wg.to(torch.float64) + warm * W0                  # warm start near the planted factor (see main())
    if normalize_w0:
        W0 = W0 / W0.sum(dim=0, keepdim=True)                  # randinit(...; normalize=true), src/interf.jl:43
    H0 = torch.rand((k, n), generator=g, dtype=torch.float64)
    Wg_d = Wg.to(device=device, dtype=tdtype)
    Hg_d = Hg[:, c0:c1].to(device=device, dtype=tdtype)
    Xt = Hg_d.t().contiguous() @ Wg_d.t().contiguous()        # (n_local, p): data generation only (rocBLAS)
    # noise: every rank draws the SAME full (n, p) field and keeps its rows, so the global X is identical for
    # every world size (strong scaling on one fixed problem); generated in row blocks to bound the temporary.
    gd = torch.Generator(device=device)
    gd.manual_seed(SEED + 1)
    blk = 2048
    for j0 in range(0, n, blk):
        j1 = min(n, j0 + blk)
        noise = torch.rand((j1 - j0, p), generator=gd, device=device, dtype=tdtype)
        lo, hi = max(j0, c0), min(j1, c1)
        if lo < hi:
            Xt[lo - c0:hi - c0].add_(noise[lo - j0:hi - j0], alpha=0.01)
    del noise
    npdt = np.float32 if tdtype == torch.float32 else np.float64
    W0h = np.asfortranarray(W0.numpy().astype(npdt))
    H0h = np.asfortranarray(H0[:, c0:c1].numpy().astype(npdt))
    return Xt, W0h, H0h


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "NMFX_BENCH_DEVICE" in os.environ:      # development aid: several ranks on one device (plumbing check on a 1-GPU box)
        local_rank = int(os.environ["NMFX_BENCH_DEVICE"])
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(self_launch(a.gpus))     # plain `python bench.py --gpus N`: become N ranks (one process per GPU)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} (or plain `python bench.py --gpus N`)")
    import nmfx
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dev_sim = os.environ.get("NMFX_BENCH_BACKEND") == "gloo-sim"   # development aid (1-GPU box): gloo rendezvous + peer-less collectives
    # development aid (1-GPU box, all ranks on NMFX_BENCH_DEVICE): gloo rendezvous + the REAL peer-window exchange between the
    # processes (RCCL refuses several ranks on one device, so there is neither an RCCL reference run nor an RCCL fallback)
    dev_gloo = os.environ.get("NMFX_BENCH_BACKEND") == "gloo-p2p"
    if dev_gloo:
        a.transport = "p2p_only"
    if world > 1:
        import torch.distributed as dist
        if dev_sim or dev_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    T = np.float32 if a.dtype == "f32" else np.float64
    tdtype = torch.float32 if a.dtype == "f32" else torch.float64
    p, n, k = a.p, a.n, a.k
    shards = a.sim_ranks if (a.sim_ranks > 1 and world == 1) else world
    c0, c1 = nmfx.dist.shard_range(n, rank, shards)
    nl = c1 - c0
    # projals: with a cold column-normalised W0 the first H = (W'W + lambda I)^-1 W'X has nearly parallel rows and H H' is not
    # numerically positive definite in fp32 (the reference's potrf! throws PosDefException on the same input), and a cold
    # un-normalised U[0,1) start leaves cond(HH' + lambda I) ~ 1e8 (tests/test_gpu_c4_c5.py): the projals workload is a warm
    # start W0 = Wg + 0.05 U, where both Grams stay at cond ~ 1e3 and the fp32 iterates are meaningful
    Xt, W0, H0 = synth(p, n, k, c0, c1, tdtype, device, normalize_w0=(a.alg != "projals"), warm=(0.05 if a.alg == "projals" else 0.0))
    torch.cuda.synchronize()

    algid = {"multmse": 0, "multdiv": 1, "projals": 2, "alspgrad": 3, "cd": 4, "greedycd": 5}[a.alg]
    eps = float(np.finfo(T).eps)
    lam = float(np.sqrt(eps)) if a.alg == "multdiv" else (float(np.cbrt(eps)) if a.alg == "projals" else 0.0)
    tiny = float(np.finfo(T).tiny)      # stop rule can never fire: exactly K iterations are executed

    def opts(iters):
        return nmfx.make_opts(T, maxiter=iters, tol=tiny, lambda_w=lam, lambda_h=lam, check_every=a.check_every,
                              maxsubiter=a.maxsubiter, precision=a.precision)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    wd = Watchdog(a.watchdog_s, rank, {"metric": "nmf_multupdate_mse_iters_per_sec" if a.alg == "multmse" else f"nmf_{a.alg}_iters_per_sec",
                                       "unit": "iters/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "higher_is_better": True,
                                       "scaling": "strong", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
                                       "config": {"workload": f"X={p}x{n} k={k} {a.dtype} alg=:{a.alg}", "parallelism": f"colshard{world}+{a.comm_mode}+{a.transport}"}})

    def make_ctx(transport, mode):
        """A context with X and the start factors resident; world > 1: a communicator of the given transport in the given mode."""
        c = nmfx.Context(T, p, nl, k, device=local_rank)
        if world > 1:                       # before set_X: attaching a communicator may change the row padding
            if dev_sim:
                c.comm_init_sim(rank, world)
                if transport != "rccl":
                    h = c.comm_p2p_export()
                    c.comm_p2p_attach([h] * world)
            else:
                nmfx.dist.init_comm(c, transport=transport)
            c.comm_set_mode(mode)
        elif shards > 1:
            c.comm_init_sim(0, shards)
            if transport != "rccl":         # the peer transport's launch sequence with every "peer" window = the own one
                h = c.comm_p2p_export()
                c.comm_p2p_attach([h] * shards)
            c.comm_set_mode(mode)
        c.set_X_device(Xt.data_ptr(), p)
        c.set_factors(W0, H0)
        return c

    def agree(ok):
        """world > 1: True only if every rank says True."""
        if world == 1:
            return ok
        import torch.distributed as dist
        tt = torch.tensor([1.0 if ok else 0.0], device=("cpu" if (dev_sim or dev_gloo) else device), dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MIN)
        return bool(tt.item() > 0.5)

    def w_hash_and_obj(c, res):
        import hashlib
        Wchk = np.empty((p, k), dtype=T, order="F")
        c.get_factors(Wchk, None)
        return hashlib.sha256(Wchk.tobytes()).hexdigest(), float(res.objvalue), bool(np.isfinite(Wchk).all())

    def probe(c, pre_err=None, iters=3):
        """A few iterations from the start factors; (ok, objective, errors): finite everywhere, W identical on every rank.  Every rank
        takes part in the exchange of results whatever happened to it before (c is None: its context could not be built)."""
        import torch.distributed as dist
        if c is None:
            mine, err = ("", float("nan"), False), pre_err or "no context"
        else:
            try:
                c.set_factors(W0, H0)
                res, _ = c.iterate(algid, opts(iters))
                mine = w_hash_and_obj(c, res)
                err = None
            except Exception as e:  # noqa: BLE001
                mine, err = ("", float("nan"), False), repr(e)
        allv = [None] * world
        dist.all_gather_object(allv, (mine, err))
        ok = (all(v[1] is None and v[0][2] and np.isfinite(v[0][1]) for v in allv) and len({v[0][0] for v in allv}) == 1
              and len({v[0][1] for v in allv}) == 1)
        return ok, allv[0][0][1], [v[1] for v in allv if v[1] is not None]

    def try_ctx(tr, md):
        try:
            return make_ctx(tr, md), None
        except Exception as e:  # noqa: BLE001
            return None, repr(e)

    # Multi-GPU: the requested transport / mode is VERIFIED before the timed region -- three iterations on it against three on plain RCCL
    # (row-sharded), same start: finite, W bit-identical on all ranks, objective equal to 1e-5 -- and replaced by the next candidate
    # when it fails (peer windows -> RCCL row-sharded -> RCCL replicated W); what ran is recorded in config.parallelism / `fallback`.
    transport, mode, fallback_log = a.transport, a.comm_mode, []
    if world > 1 and not dev_sim:
        os.environ.setdefault("NMFX_P2P_TIMEOUT_S", "20")
        ok_ref, obj_ref = False, float("nan")
        if (transport, mode) != ("rccl", "row_sharded") and not dev_gloo:
            wd.arm("verify:rccl+row_sharded")
            ref_ctx, e0 = try_ctx("rccl", "row_sharded")
            ok_ref, obj_ref, errs = probe(ref_ctx, e0)
            if ref_ctx is not None:
                ref_ctx.close()
            if not ok_ref:
                fallback_log.append({"candidate": "rccl+row_sharded (reference run)", "ok": False, "errors": errs[:2]})
        # the ladder: what was asked for first, then every other exchange the library has.  (gloo-p2p development back end: no RCCL on a
        # shared device, so the candidates are the windows' two modes)
        if dev_gloo:
            ladder = [("p2p_only", "row_sharded"), ("p2p_only", "replicated_w")]
        else:
            ladder = [("p2p", "row_sharded"), ("rccl", "row_sharded"), ("rccl", "replicated_w")] + ([("rccl", "pipelined")] if a.alg == "multmse" else [])
        cands = [(transport, mode)] + [c_ for c_ in ladder if c_ != (transport, mode)]

        def time_candidate(c):
            """--select-iters iterations of the candidate after as many untimed ones, from the start factors; ms per iteration, max over ranks
            (every rank takes part whatever happened: the iterations contain collectives)."""
            import torch.distributed as dist
            c.set_factors(W0, H0)
            c.set_final_objective(False)
            c.iterate(algid, opts(max(2, a.select_iters)))
            barrier()
            t0_ = time.perf_counter()
            c.iterate(algid, opts(max(1, a.select_iters)))
            barrier()
            tt = torch.tensor([(time.perf_counter() - t0_) / max(1, a.select_iters) * 1e3], device=("cpu" if (dev_sim or dev_gloo) else device), dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            c.set_final_objective(True)
            return float(tt.item())

        ctx = None
        verified = []                        # (ms_per_step, ladder position, context, transport, mode)
        for pos, (tr, md) in enumerate(cands):
            wd.arm(f"verify:{tr}+{md}")
            c, e0 = try_ctx(tr, md)
            ok, obj, errs = probe(c, e0)
            same = agree(ok and (not ok_ref or abs(obj - obj_ref) <= 1e-5 * abs(obj_ref)))
            entry = {"candidate": f"{tr}+{md}", "ok": same, "objective_after_3": obj,
                     "rccl_row_sharded_objective_after_3": (obj_ref if ok_ref else None), "errors": errs[:2], "ms_per_step": None}
            fallback_log.append(entry)
            if same:
                if a.select == "requested":
                    ctx, transport, mode = c, tr, md
                    break
                # timed like the timed region is (barrier, wall clock, max over ranks); a candidate that fails HERE is dropped like one
                # that failed its verification
                wd.arm(f"time:{tr}+{md}")
                try:
                    ms_c, terr = time_candidate(c), None
                except Exception as e:  # noqa: BLE001
                    ms_c, terr = float("inf"), repr(e)
                timed_ok = agree(terr is None and np.isfinite(ms_c))
                if timed_ok:
                    entry["ms_per_step"] = round(ms_c, 4)
                    verified.append((ms_c, pos, c, tr, md))
                    continue
                entry["ok"] = False
                entry["errors"] = (entry["errors"] + [terr])[:2] if terr else entry["errors"]
            if c is not None:
                c.close()
        if ctx is None and verified:
            verified.sort(key=lambda v: (v[0], v[1]))        # (the times are all-reduced: every rank sorts the same list)
            _, _, ctx, transport, mode = verified[0]
            for v in verified[1:]:
                v[2].close()
        if ctx is None:
            if rank == 0:
                print(json.dumps(dict(wd.base, value=None, status="no_working_exchange", fallback=fallback_log)), flush=True)
            raise SystemExit(4)
        ctx.set_factors(W0, H0)
        wd.arm("warmup+timed")
    else:
        ctx = make_ctx(transport, mode)

    # device pre-warm (clock ramp), outside the contract's W warm-up steps and K timed steps; every rank runs the same count
    prewarm_steps = 0
    ctx.set_final_objective(False)        # (see below: Result.objvalue is evaluated behind the timed regions)
    if a.prewarm_ms > 0:
        ctx.iterate(algid, opts(1))       # first call: lazy allocations
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.iterate(algid, opts(2))
        torch.cuda.synchronize()
        per = max(1e-6, (time.perf_counter() - t0) / 2)
        prewarm_steps = int(min(200, a.prewarm_ms * 1e-3 / per))
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([prewarm_steps], device=("cpu" if (dev_sim or dev_gloo) else device), dtype=torch.int64)
            dist.all_reduce(tt, op=dist.ReduceOp.MIN)
            prewarm_steps = int(tt.item())
        if prewarm_steps > 0:
            ctx.iterate(algid, opts(prewarm_steps))
        prewarm_steps += 3
        ctx.set_factors(W0, H0)
    if a.warmup > 0:
        ctx.iterate(algid, opts(a.warmup))
    # hipEvent pairs on the solver stream around the dominant GEMM launches and the collectives: 1 launch in 8 sampled, 1 in 4 for
    # short runs so that the roofline of a 20-step line still rests on 5 launches per kernel (1 in 2 cost 0.075 ms per iteration)
    # (ProjectedALS: 1 in 16 -- its factorisations run on a second stream whose ordering events queue behind a bracket)
    prof_mode = 0 if a.no_events else (1 if a.all_events else (4 if a.alg == "projals" else (3 if a.steps <= 32 else 2)))
    ctx.profile_enable(prof_mode)
    # Result.objvalue is evaluated AFTER the timed regions (nmfx_objective below): it is not part of an iteration -- nmf_skeleton!
    # evaluates it once per solve, behind the loop (src/common.jl:85-87) -- and one more p*n*k product inside a K-step call would
    # count as 1/K of a step (2.5 % at K = 20)
    ctx.set_final_objective(False)
    barrier()
    t0 = time.perf_counter()
    res, _ = ctx.iterate(algid, opts(a.steps))
    barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_get()
    ctx.profile_enable(0)
    # the same K steps once more WITHOUT the hipEvent brackets: what the loop costs when nobody watches (on paths with a second
    # stream -- ProjectedALS -- a bracket on the main stream delays the side stream's ordering events; both numbers go into the line)
    dt_plain = None
    if not a.no_events:
        # (from the SAME state as the bracketed region -- start factors, the same warm-up -- so that algorithms whose work depends on
        # the iterate, GreedyCD's step counts and ALSPGrad's line searches, do the same work in both regions)
        ctx.set_factors(W0, H0)
        if a.warmup > 0:
            ctx.iterate(algid, opts(a.warmup))
        barrier()
        t0 = time.perf_counter()
        res2, _ = ctx.iterate(algid, opts(a.steps))
        barrier()
        dt_plain = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([dt_plain], device=("cpu" if (dev_sim or dev_gloo) else device), dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_plain = float(tt.item())
    ctx.set_final_objective(True)
    final_objvalue = float(ctx.objective(algid, opts(1)))     # of the factors the last timed region left (collective when sharded)
    wd.disarm()
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=("cpu" if (dev_sim or dev_gloo) else device), dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert res.niters == a.steps, (res.niters, a.steps)
    # N > 1: the ranks must hold the SAME W (bit for bit) and report the same objective -- the exchange step is the only thing
    # that keeps them together, so this is the on-hardware check of the RCCL path (outside the timed region)
    consistency = None
    if world > 1:
        import hashlib
        import torch.distributed as dist
        Wchk = np.empty((p, k), dtype=T, order="F")
        ctx.get_factors(Wchk, None)
        mine = (hashlib.sha256(Wchk.tobytes()).hexdigest(), final_objvalue, bool(np.isfinite(Wchk).all()))
        allv = [None] * world
        dist.all_gather_object(allv, mine)
        consistency = {"W_identical_on_all_ranks": len({v[0] for v in allv}) == 1,
                       "objective_identical_on_all_ranks": len({v[1] for v in allv}) == 1,
                       "finite": all(v[2] for v in allv) and bool(np.isfinite(final_objvalue))}

    if rank == 0:
        ms = dt / a.steps * 1e3
        if a.alg in ("multmse", "cd", "greedycd"):
            # cd / greedycd: the same two big GEMMs and Grams; the row sweeps add 2k^2(p+n) (cd) or a data-dependent
            # number of greedy steps (reported below), both priced like the multmse update products
            f_alg = 4.0 * p * n * k + 4.0 * k * k * (p + n)
        elif a.alg == "multdiv":
            f_alg = 8.0 * p * n * k
        elif a.alg == "projals":
            f_alg = 4.0 * p * n * k + 2.0 * k * k * (p + n) + 2.0 * k * k * n + 2.0 * p * k * k + float(k) ** 3
        else:
            # alspgrad: data-dependent: the two big GEMMs and Grams, plus one k x k GEMM per executed
            # inner iteration and per executed back-tracking step; the split between the H side (2 k^2 n each) and the
            # W side (2 p k^2 each) is not reported by the device counters, so both are priced at the mean of the two
            per_small = float(k) * k * (p + n)
            f_alg = 4.0 * p * n * k + 2.0 * k * k * (p + n) + per_small * (res.inner_iters + res.backtracks) / a.steps
        sim_div = float(shards) if shards != world else 1.0
        # dominant kernel = the GEMM family with the largest summed time
        gemms = [s for s in prof if s["flops"] > 0]
        dom = max(gemms, key=lambda s: s["ms_total"]) if gemms else None
        roof = None
        if dom is not None:
            avg_s = dom["ms_total"] / dom["launches"] * 1e-3
            ach = dom["flops"] / dom["launches"] / avg_s / 1e12
            traffic, traffic_source = None, None
            tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            default_workload = (a.alg == "multmse" and a.dtype == "f32" and (p, n, k) == (16384, 16384, 256) and world == 1 and shards == 1
                                and a.precision == "fp32")
            if default_workload and a.traffic == "live":
                traffic, traffic_source = live_traffic(dom["name"])
            if default_workload and traffic is None and a.traffic != "none" and os.path.exists(tf):   # the committed PMC passes of this workload
                try:
                    traffic = json.load(open(tf)).get(dom["name"])
                    traffic_source = ("static: profiles/pmc_traffic.json (rocprofv3 --pmc passes of this kernel on this workload, taken when the "
                                      "profile was committed; not re-measured in this run)")
                except Exception:
                    traffic = None
            # bf16x3: three dense-bf16 MFMA products (2.5 PFLOP/s peak) per fp32-equivalent product
            on_bf16 = a.precision == "bf16x3" and a.dtype == "f32" and ("bf16x3" in dom["name"] or dom["name"].startswith("gemm_WH_"))
            peak = (2500.0 / 3.0) if on_bf16 else (PEAK_FP32_MFMA_TFLOPS if a.dtype == "f32" else 78.6)
            roof = {"bound": "mfma", "kernel": dom["name"], "achieved": round(ach, 2),
                    "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4),
                    "traffic": traffic, "traffic_source": traffic_source,
                    "flops_per_launch": dom["flops"] / dom["launches"],
                    "avg_launch_ms": round(avg_s * 1e3, 4), "launches": dom["launches"]}
        out = {
            "metric": "nmf_multupdate_mse_iters_per_sec" if a.alg == "multmse" else f"nmf_{a.alg}_iters_per_sec",
            "value": round(a.steps / dt, 4), "unit": "iters/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": a.dtype if a.precision == "fp32" else f"{a.dtype} (big GEMMs: 3 x bf16 MFMA products, fp32 accumulate; opt-in)",
            "data": "synthetic",
            "config": {"workload": f"X={p}x{n} k={k} {a.dtype} alg=:{a.alg} (planted-rank dense X, seed {SEED}), "
                                   f"column-sharded over {world} GPU(s)", "p": p, "n": n, "k": k,
                       "parallelism": f"colshard{world}" + (f"+{mode}+{transport}" if (world > 1 or shards > 1) else ""), "precision": a.precision,
                       "check_every": (a.check_every if a.check_every < (1 << 30) else "never (stop rule evaluated on the device only)")},
            # `value` / `ms_per_step` are the region with the sampled hipEvent brackets (the roofline's launch times come from it);
            # the same K steps without any bracket right behind it:
            "ms_per_step_no_events": (round(dt_plain / a.steps * 1e3, 4) if dt_plain is not None else round(ms, 4)),
            "prewarm_steps": prewarm_steps,   # untimed, before the W warm-up steps (clock ramp; --prewarm-ms)
            "event_brackets": {0: "none", 1: "every launch", 2: "1 launch in 8 of the dominant GEMMs", 3: "1 launch in 4 of the dominant GEMMs",
                               4: "1 launch in 16 of the dominant GEMMs"}[prof_mode],
            # (--sim-ranks G times ONE rank of G on one GPU: that rank's share of the flops, so the fraction stays a fraction)
            "gflops_algorithmic": round(f_alg / sim_div * a.steps / dt / 1e9, 1),
            "frac_of_mfma_peak": round(f_alg / sim_div * a.steps / dt / 1e12 / (((2500.0 / 3.0) if a.precision == "bf16x3" else
                                                                                  (PEAK_FP32_MFMA_TFLOPS if a.dtype == "f32" else 78.6)) * world), 4),
            "objvalue": final_objvalue,   # evaluated after the timed region (see above)
            "roofline": roof,
            # per kernel: average launch time; algorithmic TFLOP/s and algorithmic HBM GB/s where the launch site states them
            "kernels": [dict({"name": s["name"], "launches": s["launches"], "avg_us": round(s["ms_total"] / s["launches"] * 1e3, 2)},
                             **({"tflops": round(s["flops"] / s["ms_total"] / 1e9, 1)} if s["flops"] > 0 else {}),
                             **({"hbm_gbs": round(s["bytes"] / s["ms_total"] / 1e6, 1)} if s["bytes"] > 0 else {})) for s in prof],
        }
        # the bandwidth-bound passes of the divergence / objective path against the HBM roofline (north_star: "achieved HBM
        # GB/s for the bandwidth-bound divergence/objective passes"): algorithmic bytes = X read + Q written (ratio pass),
        # X read (objective pass) -- SURVEY.md section 8d -- over the launch time
        hb = [s for s in prof if s["bytes"] > 0 and s["name"] in ("gemm_WH_ratio", "gemm_WH_sqdist", "gemm_WH_kldiv")]
        if hb:
            out["roofline_hbm"] = [{"bound": "hbm", "kernel": s["name"], "achieved": round(s["bytes"] / s["ms_total"] / 1e6, 1),
                                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(s["bytes"] / s["ms_total"] / 1e6 / PEAK_HBM_GBS, 4),
                                    "bytes_per_launch": s["bytes"] / s["launches"], "avg_launch_ms": round(s["ms_total"] / s["launches"], 4),
                                    "note": "fused into the W*H MFMA GEMM (2pnk flop per launch): the launch is MFMA-bound, so the HBM "
                                            "fraction is what the fusion leaves unused, not a shortfall"} for s in hb]
        if consistency is not None:
            out["multi_gpu_consistency"] = consistency
        if fallback_log:
            out["exchange_verification"] = {"requested": f"{a.transport}+{a.comm_mode}", "ran": f"{transport}+{mode}",
                                            "selected_by": ("fastest verified candidate (candidates[*].ms_per_step: %d iterations after as many untimed ones, "
                                                            "max over ranks)" % a.select_iters) if a.select == "fastest" else "first verified candidate in ladder order",
                                            "candidates": fallback_log}
        # the exchange step's collectives (hipEvent brackets on the solver's stream: includes waiting for the slowest peer)
        coll = [s for s in prof if s["name"].startswith("comm_")]
        if coll:
            out["collectives"] = {"transport": (f"sim of {transport} (device-local)" if (dev_sim or shards != world) else transport), "mode": mode,
                                  "per_iteration_us": round(sum(s["ms_total"] / s["launches"] for s in coll) * 1e3, 1),
                                  "calls": [{"name": s["name"], "avg_us": round(s["ms_total"] / s["launches"] * 1e3, 1), "sampled": s["launches"],
                                             "bytes": s["bytes"] / s["launches"]} for s in coll]}
        if shards != world:
            out["sim_ranks"] = shards
            out["metric"] += f"_SIMULATED_rank0_of_{shards}_compute_only"
        if world == 1 and shards == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = (cpu_baseline(p, n, k, T, Xt, W0, H0, a.cpu_sample_cols, a.cpu_full_iters) if a.alg == "multmse" else
                                   cpu_baseline_generic(a.alg, p, n, k, T, Xt, W0, H0, a.cpu_sample_cols, lam, a.maxsubiter))
        if a.alg == "greedycd":
            out["greedy_steps_per_step"] = res.inner_iters / a.steps
        if a.alg == "alspgrad":
            out["config"]["maxsubiter"] = a.maxsubiter
            out["inner_iters_per_step"] = res.inner_iters / a.steps
            out["backtracks_per_step"] = res.backtracks / a.steps
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` typed as it stands (no launcher): re-run this script as N ranks under torch.distributed.run on this
    node -- rendezvous on 127.0.0.1 at a free port, the caller's arguments unchanged -- and hand back its exit code.  Rank 0 of the
    child job prints the one JSON line on the inherited stdout.  With the development back ends (NMFX_BENCH_BACKEND=gloo-p2p / gloo-sim)
    on a box with fewer than N devices every rank is placed on device 0."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if env.get("NMFX_BENCH_BACKEND") in ("gloo-p2p", "gloo-sim") and "NMFX_BENCH_DEVICE" not in env and torch.cuda.device_count() < n:
        env["NMFX_BENCH_DEVICE"] = "0"
    # the caller's arguments travel in the environment: torch.distributed.run's own parser trips over script options that are
    # prefixes of its own (`--n` "could match --nnodes, --nproc-per-node, ...") even behind the script name
    env["NMFX_BENCH_ARGV"] = json.dumps(sys.argv[1:])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)]
    return subprocess.call(cmd, env=env)


def live_traffic(kernel):
    """HBM bytes per launch of the dominant kernel, measured NOW: two rocprofv3 passes (--pmc FETCH_SIZE, then --pmc WRITE_SIZE; counters in
    runs of their own with --kernel-trace only, as MI355X_MICROARCH.md prescribes) of this script on the same workload (4 steps), and
    traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are KiB, and on gfx950 FETCH_SIZE reports half of the bytes of a wide
    coalesced read (the guide's correction; calibrated in profiles/r0*_bench_multmse_c3.md on the split-K combine, whose byte count is known).
    Returns (bytes or None, source string)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, None
    # the launches behind the bench names (csrc/solver.hpp): W'X = both operands contraction-contiguous, XH' = both strided; 128 x 128 tiles
    # (round 5: X*H' runs on the transposed images of X and H, i.e. on the SAME instantiation as W'X -- both operands contraction-contiguous;
    # the two launches read X once each plus a 16 MiB factor and write two 16 MiB slabs: the counters are averaged over both)
    pat = {"gemm_WtX": "gemm_mfma_kernel<float, 0, 0, 128, 128, 2, 2, nmfx::EpiStore<float>", "gemm_XHt": "gemm_mfma_kernel<float, 0, 0, 128, 128, 2, 2, nmfx::EpiStore<float>"}.get(kernel)
    if pat is None:
        return None, None
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", td, "--", sys.executable, os.path.abspath(__file__), "--steps", "4",
                       "--warmup", "2", "--no-cpu-baseline", "--no-events", "--traffic", "none"]
                subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
                tot, cnt = 0.0, 0
                for f in glob.glob(td + "/**/*counter_collection.csv", recursive=True):
                    for r in csv.DictReader(open(f)):
                        if r.get("Counter_Name") == counter and pat in r.get("Kernel_Name", ""):
                            tot += float(r["Counter_Value"])
                            cnt += 1
                if cnt == 0:
                    return None, None
                vals[counter] = (tot / cnt, cnt)
    except Exception:  # noqa: BLE001
        return None, None
    bytes_ = (2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024.0
    return round(bytes_), (f"live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two runs of this script behind the timed region (4 steps each; "
                           f"{vals['FETCH_SIZE'][1]} / {vals['WRITE_SIZE'][1]} dispatches of the kernel), (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
                           f"with the gfx950 FETCH_SIZE correction; raw KiB {vals['FETCH_SIZE'][0]:.0f} / {vals['WRITE_SIZE'][0]:.0f}")


def _blas_pool():
    """(cap, describe): the BLAS thread pool NumPy will use -- the wheel's OpenBLAS has a compile-time cap that can be far below
    the host's core count, and several BLAS / OpenMP runtimes may be loaded (torch brings its own)."""
    try:
        from threadpoolctl import threadpool_info
        info = threadpool_info()
    except Exception:  # noqa: BLE001
        return None, None
    np_blas = [d for d in info if d.get("user_api") == "blas" and "numpy" in (d.get("filepath") or "")] or \
              [d for d in info if d.get("user_api") == "blas"]
    cap = max((d.get("num_threads") or 0) for d in np_blas) if np_blas else None
    desc = [{"api": d.get("internal_api"), "version": d.get("version"), "threads": d.get("num_threads"),
             "lib": os.path.basename(d.get("filepath") or "")} for d in info]
    return cap, desc


def cpu_baseline(p, n, k, T, Xt, W0, H0, ns, full_iters=2):
    """NMF.jl's CPU path = the oracle's restatement of the reference's operation sequence with prepare_state's arrays allocated
    ONCE (oracle/nmf_oracle.py::_MultMSEState, like MultUpdMSE_State, src/multupd.jl:63-80), on the first `ns` columns of the same X
    (per-iteration cost is linear in n); value is scaled to the full problem.
    PURE iterations are timed: update_wh! + the preW/preH copies + stop_condition of nmf_skeleton! (src/common.jl:66-73),
    i.e. what one GPU 'step' does -- not prepare_state's W*H product, not the final objective pass.
    It runs in a PROCESS OF ITS OWN (oracle/cpu_time.py): this process carries torch's OpenMP pool and a second OpenBLAS next to
    NumPy's; the child loads NumPy's OpenBLAS only.  BLAS pool pinned per trial (threadpoolctl) at the pool's cap and at 1/2 ... 1/16
    of it (skinny k = 256 products do not scale to every core; two sockets); the FASTEST setting is the baseline, `cores` = the
    threads it used, and the line carries the seconds of every call site (each mul!, each element-wise loop, the copies,
    stop_condition) of that setting."""
    import subprocess
    import tempfile
    ns = min(ns, n)
    Xs = np.asfortranarray(Xt[:ns, :].cpu().numpy().T)            # p x ns, column-major
    host_cores = os.cpu_count()
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "sample.npz")
        np.savez(f, X=Xs, W0=np.asfortranarray(W0), H0=np.asfortranarray(H0[:, :ns]))
        env = {k_: v for k_, v in os.environ.items() if k_ not in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_time.py"), f, "4.0", str(n)], capture_output=True, text=True, timeout=900, env=env)
    if r.returncode != 0:
        return {"value": None, "unit": "iters/s", "cores": None, "host_cores": host_cores, "kind": "port", "error": (r.stderr or r.stdout)[-400:]}
    d = json.loads(r.stdout.strip().splitlines()[-1])
    best = d["best"]
    # per-iteration cost of the full problem: everything that touches X, WH or H scales with n / ns; the p x k passes over W do not
    t_iter_full = best.get("seconds_per_iter_full_est", best["seconds_per_sample_iter"] * (n / ns))
    used = best["blas_threads"]
    out = {"value": round(1.0 / t_iter_full, 5), "unit": "iters/s", "cores": used if used else host_cores, "host_cores": host_cores,
           "kind": "port",
           "sample": f"first {ns} of {n} columns of the same X = {ns / n:.4f} of the workload (p={p}, k={k}); {best['iters']} timed pure outer "
                     f"iterations (update_wh! + preW/preH copies + stop_condition; prepare_state and the final objective are not "
                     f"timed) of oracle/nmf_oracle.py::_MultMSEState -- the reference's 6-mul! sequence with its state allocated once, in a "
                     f"process of its own (NumPy's OpenBLAS the only BLAS loaded) pinned to {used} threads (pool cap {d['pool_cap']} on a "
                     f"{host_cores}-core host; fastest of `thread_trials`), element-wise loops single-threaded like stock Julia "
                     f"(stop_condition's sums by np.sum, not by the oracle's serial-recurrence emulation); the column-dependent phases scaled by n/{ns}, the "
                     f"p x k passes over W counted once",
           "seconds_per_iter_full_est": round(t_iter_full, 4),
           "gflops_reference_equiv": round(12.0 * p * n * k / t_iter_full / 1e9, 1),
           # where the sample iteration's time goes at the fastest setting (seconds per call site), and the six mul! on their own
           "phase_seconds": best["phase_seconds"], "gemm_phases_gflops": best["gemm_gflops"], "gemm_seconds": best["gemm_seconds"],
           "non_gemm_seconds": best["non_gemm_seconds"], "column_independent_seconds": best.get("column_independent_seconds"),
           "thread_trials": [{k_: t[k_] for k_ in ("blas_threads", "iters", "seconds_per_sample_iter", "gemm_gflops", "non_gemm_seconds")} for t in d["thread_trials"]],
           "blas": d["blas"]}
    # ... and the FULL problem, measured (SURVEY.md section 8d "time >= 3 iterations at each config"): the same child process on all n
    # columns -- X handed over as an uncompressed .npy, the state pre-touched by prepare_state and one warm-up iteration, the BLAS pool
    # pinned to the sample's fastest setting, `full_iters` timed iterations.  When it succeeds THIS is `value`; the scaled sample stays
    # beside it as `scaled_sample_value`.
    out["measured_full_size"] = False
    if full_iters > 0 and ns < n:
        full = cpu_full_size(p, n, k, Xt, W0, H0, used, full_iters)
        out["full_size_run"] = full
        if full.get("seconds_per_iter") is not None:
            out["scaled_sample_value"] = out["value"]
            out["value"] = round(1.0 / full["seconds_per_iter"], 5)
            out["measured_full_size"] = True
            out["gflops_reference_equiv"] = round(12.0 * p * n * k / full["seconds_per_iter"] / 1e9, 1)
            out["sample"] = (f"FULL problem, measured: all {n} columns of the same X (p={p}, k={k}), {full['iters']} timed pure outer iterations (update_wh! + "
                             f"preW/preH copies + stop_condition) of oracle/nmf_oracle.py::_MultMSEState after prepare_state and one untimed warm-up "
                             f"iteration, own process, BLAS pool pinned to {used} threads (the fastest setting of the column sample).  The scaled "
                             f"column sample (`scaled_sample_value`): " + out["sample"])
    elif ns >= n:
        out["measured_full_size"] = True      # the sample IS the problem
    out["julia_reference"] = julia_reference(p, ns, k, T)
    return out


def cpu_full_size(p, n, k, Xt, W0, H0, threads, iters):
    """`iters` full-size iterations of the CPU port in a process of its own; {"seconds_per_iter": ..., ...} or {"error": ...}."""
    import subprocess
    import tempfile
    t_all = time.perf_counter()
    try:
        tmp_root = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        with tempfile.TemporaryDirectory(dir=tmp_root) as td:
            Xh = Xt.cpu().numpy()                                   # (n, p) row-major == p x n column-major
            np.save(os.path.join(td, "X.npy"), Xh.T)                # Fortran-order .npy of the p x n matrix (no copy: .T of a C array)
            del Xh
            np.savez(os.path.join(td, "f.npz"), W0=np.asfortranarray(W0), H0=np.asfortranarray(H0))
            env = {k_: v for k_, v in os.environ.items() if k_ not in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
            cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_time.py"), os.path.join(td, "f.npz"), "0.0", str(n), "--x-npy", os.path.join(td, "X.npy"),
                   "--min-iters", str(iters), "--max-iters", str(iters)] + (["--threads", str(threads)] if threads else [])
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        if r.returncode != 0:
            return {"seconds_per_iter": None, "error": (r.stderr or r.stdout)[-400:]}
        d = json.loads(r.stdout.strip().splitlines()[-1])
        b = d["best"]
        return {"seconds_per_iter": b["seconds_per_sample_iter"], "iters": b["iters"], "blas_threads": b["blas_threads"], "columns": d["ns"],
                "gemm_seconds": b["gemm_seconds"], "gemm_gflops": b["gemm_gflops"], "non_gemm_seconds": b["non_gemm_seconds"],
                "phase_seconds": b["phase_seconds"], "wall_seconds_incl_setup": round(time.perf_counter() - t_all, 1)}
    except Exception as e:  # noqa: BLE001
        return {"seconds_per_iter": None, "error": repr(e)}


def cpu_baseline_generic(alg, p, n, k, T, Xt, W0, H0, ns, lam, maxsubiter):
    """The other algorithms: NMF.solve! of the NumPy restatement on the first `ns` columns, timed as (3 iterations) - (1 iteration)
    so that prepare_state and the final objective cancel; scaled by n / ns.  A slow first run (> 20 s) is reported as it is."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nmf_oracle as orc
    ns = min(ns, n)
    Xs = np.asfortranarray(Xt[:ns, :].cpu().numpy().T)
    tiny = float(np.finfo(T).tiny)

    def run(iters):
        Ws, Hs = W0.copy(order="F"), np.asfortranarray(H0[:, :ns].copy())
        o = orc.Opts(maxiter=iters, tol=tiny, lambda_w=lam, lambda_h=lam)
        if hasattr(o, "maxsubiter"):
            o.maxsubiter = maxsubiter
        t0 = time.perf_counter()
        orc.solve(alg, Xs, Ws, Hs, o)
        return time.perf_counter() - t0

    t1 = run(1)
    if t1 > 20.0:
        t_iter, how = t1, "ONE outer iteration incl. prepare_state and the final objective (a second run would exceed the time bound)"
    else:
        t3 = run(3)
        t_iter, how = max((t3 - t1) / 2.0, 1e-9), "(3 iterations) - (1 iteration) over 2: prepare_state and the final objective cancel"
    t_full = t_iter * (n / ns)
    cap, blas = _blas_pool()
    return {"value": round(1.0 / t_full, 5), "unit": "iters/s", "cores": cap if cap else os.cpu_count(), "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"first {ns} of {n} columns of the same X = {ns / n:.4f} of the workload (p={p}, k={k}); NMF.solve!({alg}) of "
                      f"oracle/nmf_oracle.py, {how}; NumPy's OpenBLAS at its pool cap of {cap} threads on a {os.cpu_count()}-core host; "
                      f"time scaled by n/{ns}", "blas": blas,
            "seconds_per_iter_full_est": round(t_full, 3)}


def julia_reference(p, ns, k, T):
    """if a `julia` with NMF.jl happens to be installed on the box, ALSO time the real reference
    (NMF.solve! on a rand(p, ns) matrix of the sample's shape) and report it, labelled as such.  Neither this image nor the
    GPU box ships Julia, so this normally reports found = false."""
    import shutil
    import subprocess
    exe = shutil.which("julia")
    if exe is None:
        return {"found": False}
    jt = "Float32" if T == np.float32 else "Float64"
    prog = (f"using NMF, Random; Random.seed!(1); X = rand({jt}, {p}, {ns}); W = rand({jt}, {p}, {k}); H = rand({jt}, {k}, {ns}); "
            f"alg = NMF.MultUpdate{{{jt}}}(obj=:mse, maxiter=2, tol=floatmin({jt})); NMF.solve!(alg, X, copy(W), copy(H)); "
            f"alg = NMF.MultUpdate{{{jt}}}(obj=:mse, maxiter=4, tol=floatmin({jt})); t = @elapsed NMF.solve!(alg, X, W, H); "
            f"println(\"NMFJL_SECONDS_PER_ITER=\", t / 4, \" THREADS=\", Threads.nthreads())")
    try:
        r = subprocess.run([exe, "-e", prog], capture_output=True, text=True, timeout=300)
        for line in r.stdout.splitlines():
            if line.startswith("NMFJL_SECONDS_PER_ITER="):
                return {"found": True, "kind": "reference", "raw": line.strip(),
                        "note": f"NMF.solve!(MultUpdate(obj=:mse)) on rand({p},{ns}), 4 iterations incl. prepare_state and the final objective"}
        return {"found": True, "error": (r.stderr or r.stdout)[-300:]}
    except Exception as e:  # noqa: BLE001
        return {"found": True, "error": repr(e)}


if __name__ == "__main__":
    main()
