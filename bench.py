#!/usr/bin/env python3
"""Benchmark of the discrete graph-DDPM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment: this process makes no GPU call, starts the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`),
relays rank 0's JSON line and exits with the launcher's status.  Under torch.distributed.run (RANK / LOCAL_RANK /
WORLD_SIZE set) it is one rank of the job.

Workload (default = BASELINE.json configs[3], the configuration the metric is quoted on): synthetic SBM graphs with N = 200
nodes, the 8 x 128 PPGN of gridsearch_ppgn_simple.py:19, `ppgn_simple` sampling with 64 noise levels, 32 graphs per GPU.
One step = one full reverse chain (x_T init, then 64 x [re-noise, denoiser forward, reverse step]) over one batch; graphs
are independent, so ranks take disjoint graph ids and exchange nothing (weak scaling).  `value` = graphs finished per
second over all ranks, denoiser in `--precision` (default bf16x3: float32 tensors, H = 128 matrix products as three bf16
MFMAs on split operands with float32 accumulation, held to the float32 parity bar by tests/test_gpu_ppgn.py).  `exact_f32`
carries the same chains over the same number of steps with exact-float32 MFMAs and its own roofline object.  The line also
carries the training throughput (optimizer steps/s at global batch 32 x N GPUs, data parallel with one flat-gradient
all-reduce per step), a per-kernel-class table measured live with HIP events on the launch stream, and the CPU baseline.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0   # idem, dense bf16
HBM_PEAK_GBS = 8000.0

# kernel classes x arithmetic of dppm_prof_end (include/dppm_hip_debug.h) -> the kernels that run in them
KERNEL_NAMES = {
    ("wstat_mlp2", "bf16x3"): "w4_mlp_kernel<3,...> (both per-position MLPs of a PPGN layer, model/ppgn.py:58-59; float32 in HBM, operands split into two "
                              "bf16 pieces in LDS, three v_mfma_f32_16x16x32_bf16 per product; reads X, writes o1 and o2)",
    ("wstat_mlp2", "bf16"): "w4_mlp_kernel<1,...> (the MLP pair with bf16-rounded operands, one MFMA per product)",
    ("wstat_mlp2", "f32"): "ws_mlp_kernel (both per-position MLPs of a PPGN layer, v_mfma_f32_16x16x4_f32; training: hidden activations saved; "
                           "backward form: dpre and dX)",
    ("wstat_sum2", "bf16x3"): "w4_lin2_kernel<3,...> (skip-concat Linear m4, model/ppgn.py:66-69, with the masked statistics of the fused instance norm)",
    ("wstat_sum2", "bf16"): "w4_lin2_kernel<1,...>",
    ("wstat_sum2", "f32"): "ws_lin_kernel<2> (skip-concat Linear m4 and the layer_cat_lin pairs, v_mfma_f32_16x16x4_f32)",
    ("wstat_cols", "bf16x3"): "w3_lin_kernel<3,1> (after_conv, model/ppgn.py:331)",
    ("wstat_cols", "f32"): "ws_lin_kernel<1> (after_conv; transposed data-gradient forms), exact f32",
    ("wstat_cols", "f32x6"): "w6_lin_kernel<1,TRANS> (after_conv forward and every transposed data-gradient form: three bf16 pieces per operand, six "
                             "v_mfma_f32_16x16x32_bf16 per product)",
    ("wstat_mlp2", "f32x6"): "w6_lin_kernel pair of the MLP backward (dpre = (W2^T dOut) LeakyReLU'(hid), then dX += W1^T dpre), three-piece split",
    ("plane_matmul", "bf16x3"): "plane4_kernel<NT,3,...> (per-channel N x N contraction, model/ppgn.py:62; two bf16 pieces per operand, three "
                                "v_mfma_f32_16x16x32_bf16 per product)",
    ("plane_matmul", "bf16"): "plane4_kernel<NT,1,...>",
    ("plane_matmul", "f32x6"): "plane4_kernel<NT,6,1,TA,TB> (the contraction and its two gradient forms in training: three bf16 pieces, six products)",
    ("plane_matmul", "f32"): "plane_mm16_kernel / plane_mm_kernel (per-channel N x N contraction on v_mfma_f32_16x16x4_f32)",
    ("inorm", "none"): "inorm_fwd_kernel / inorm_bwd_kernel (masked instance norm, util/model_helper.py:176-191)",
    ("linear_gemm", "f32"): "gemm_f32_kernel / mlp32_*_kernel / lin32t_kernel (per-position linears of networks with H != 128, exact f32 MFMA)",
    ("wgrad_gemm", "f32x6"): "ws6_wgrad_kernel (128 x 128 weight gradients over all positions, three-piece bf16 split, six products)",
    ("wgrad_gemm", "f32"): "ws_wgrad_kernel / wgrad32m_kernel / generic split-K GEMM (weight gradients, exact f32 MFMA)",
    ("head", "bf16x3"): "head3_kernel<3,...> (layer_cat_lin + after_conv + final_lin in one pass, model/ppgn.py:318-333)",
    ("head", "bf16"): "head3_kernel<1,...>",
}
BF16_PRODUCTS = {"bf16x3": 3, "bf16": 1, "f32x6": 6}    # bf16 MFMAs per float32 product


def sbm_graphs(rng, B, N, p_intra=0.3, p_inter=0.005):
    """k = ceil(N/25) blocks, all nodes active, symmetric, zero diagonal."""
    k = -(-N // 25)
    A = np.zeros((B, N, N), np.float32)
    for b in range(B):
        blocks = rng.integers(0, k, size=N)
        same = blocks[:, None] == blocks[None, :]
        up = np.triu(rng.random((N, N)) < np.where(same, p_intra, p_inter), 1).astype(np.float32)
        for i in range(N - 1):      # keep every node at degree >= 1 so flags = adj.sum(-1) > 0 marks all N active
            up[i, i + 1] = 1.0
        A[b] = up + up.T
    return A


def fwd_flops_per_graph(N, H, L):
    """SURVEY.md section 8(d)."""
    return 4 * N * N * H + L * (12 * N * N * H * H + 2 * H * N ** 3) + 2 * N * N * H * (L * H + 2) + 2 * N * N * H * H + 2 * N * N * H


# --------------------------------------------------------------------------------------------------- CPU baseline
def _cpu_leg(N, H, L, T, Bc, faithful, threads, budget_s, max_steps=4):
    """Time chain steps of the oracle's sampler on `threads` host threads.  faithful: the mask is rebuilt every step with the
    reference's Python triple loop (utils/graph_utils.py:312-319, called from take_step at sample_ppgn_simple.py:56) and the
    p = 1/2 matrix of gen_init_data with its Python loop (:32-37, once per chain, amortised over the T steps); otherwise the
    vectorised restatement.  One warm-up step, then the median of up to `max_steps` (>= 3) timed steps."""
    from oracle import diffusion as od, ppgn as op
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        rng = np.random.default_rng(1234)
        A = sbm_graphs(rng, Bc, N)
        flags = (A.sum(-1) > 1e-5).astype(np.float32)
        W = op.init_weights(L, H, seed=1234)
        t_init = 0.0
        if faithful:
            t0 = time.perf_counter()
            od.gen_init_prob_loop(flags)
            t_init = time.perf_counter() - t0
        x = od.gen_init_data(flags, rng.random((Bc, N, N), dtype=np.float32))
        sig = od.schedule_simple(T)
        times = []
        t_begin = time.perf_counter()
        with torch.no_grad():
            for c in range(1 + max_steps):
                s = sig[len(sig) - c - 1]
                t0 = time.perf_counter()
                mask = od.generate_mask_loop(flags) if faithful else od.generate_mask(flags)
                x, _ = od.take_step_simple(x, lambda a: op.powerful_forward(W, a, mask, s, L).squeeze(-1), flags, s,
                                           rng.random((Bc, N, N), dtype=np.float32), rng.random((Bc, N, N), dtype=np.float32))
                if c > 0:
                    times.append(time.perf_counter() - t0)
                if len(times) >= 3 and time.perf_counter() - t_begin > budget_s:
                    break
        step = float(np.median(times))
        per_graph_chain = (step * T + t_init) / Bc
        return {"graphs_per_s": 1.0 / per_graph_chain, "s_per_chain_step": step, "graphs": Bc, "timed_steps": len(times), "threads": threads}
    finally:
        torch.set_num_threads(old)


def cpu_baseline(N, H, L, T):
    """The oracle (oracle/ppgn.py + oracle/diffusion.py, torch CPU fp32) on a bounded sample of the same workload (SURVEY.md
    section 8d): the reference-faithful form (whole-batch forward as the reference's sampler calls it, mask and p = 1/2 matrix
    built by the reference's Python loops) and the batched restatement (vectorised mask), each on 1 thread and on all the
    threads torch uses on this host; extrapolated from a few chain steps to the full T-step chain."""
    k_all = torch.get_num_threads()
    big = N >= 100
    Bc = 1 if big else 4
    legs = {}
    for name, faithful in (("reference_faithful", True), ("batched_restatement", False)):
        for k in sorted({1, k_all}):
            legs[f"{name}_k{k}"] = _cpu_leg(N, H, L, T, Bc, faithful, k, budget_s=8.0 if big else 4.0)
    best = legs[f"reference_faithful_k{k_all}"]
    return {"value": best["graphs_per_s"], "unit": "graphs/s", "cores": k_all, "kind": "port",
            "sample": f"{Bc} graph(s) x 1 warm-up + {best['timed_steps']} timed chain steps of {T} (median), N={N}, {L}x{H} PPGN, oracle "
                      f"(torch CPU fp32) in the reference-faithful form on {k_all} threads, extrapolated to the full chain",
            "legs": legs}


# --------------------------------------------------------------------------------------------------- launcher
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """Parent of an N-rank run: no GPU call is made here."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n, 1))))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:                 # relay; the one JSON line of rank 0 is re-printed last, alone
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("bench.py: the ranks exited without printing a result line", file=sys.stderr)
        rc = 1
    return rc


# --------------------------------------------------------------------------------------------------- roofline objects
def kernel_table(prof, dt, traffic_json, mode):
    """Per (kernel class, arithmetic): launches, average launch time, share of the timed region and a roofline against the roof
    that bounds THAT arithmetic.  The library tags every timed launch with the arithmetic it ran in, so a class never mixes pipes:
      - exact-f32 MFMA launches ("f32"): bound = the f32 matrix pipe, algorithmic FLOP/s against 157.3 TFLOP/s;
      - launches on the bf16 pipe with split operands ("bf16x3", "bf16", "f32x6") and the instance norm ("none"): bound = HBM,
        algorithmic bytes per launch (every distinct operand read once, every result written once, counted by the library) / average
        launch time against 8 TB/s; their matrix work is shown beside it as bf16_mfma_frac = (3, 1 or 6 bf16 MFMAs per product) x
        algorithmic FLOP/s / 2.5 PFLOP/s.  No bf16-pipe work is ever divided by the f32 peak.
    Keys are "class/arith".  `traffic` (PMC bytes per launch) applies to the classes of the sampling run in `mode`."""
    table = {}
    for (k, ar), c in prof.items():
        if c["launches"] == 0:
            continue
        avg_ms = c["ms"] / c["launches"]
        flops = 0.0 if ar == "none" else c["work"]
        tfl = flops / (c["ms"] * 1e-3) / 1e12 if c["ms"] > 0 else 0.0
        gbs = c["bytes"] / (c["ms"] * 1e-3) / 1e9 if c["ms"] > 0 else 0.0
        e = {"kernel": KERNEL_NAMES.get((k, ar), k), "arithmetic": ar, "launches": c["launches"], "avg_launch_ms": avg_ms,
             "share_of_step_time": c["ms"] * 1e-3 / dt, "bytes_per_launch": c["bytes"] / c["launches"], "flops_per_launch": flops / c["launches"]}
        if ar == "f32":
            e.update({"bound": "mfma", "achieved": tfl, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / F32_MFMA_PEAK_TFLOPS,
                      "hbm_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS})
        else:
            e.update({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS})
            if ar in BF16_PRODUCTS:
                e["f32_equivalent_tflops"] = tfl
                e["bf16_mfma_frac"] = BF16_PRODUCTS[ar] * tfl / BF16_MFMA_PEAK_TFLOPS
        e["traffic"] = (traffic_json or {}).get(k) if ar in (mode, "none") else None
        table[f"{k}/{ar}"] = e
    return table


def path_object(table, N, H, L, B, levels_timed, dt_level_s, mode):
    """The whole denoiser level against SURVEY.md section 8(d): HBM bytes the kernels of one level move by the library's own count
    (in plane-sets S = B*H*N*ld*4 bytes) against the algorithmic bytes of a fully fused path, 4*N^2*(2 + H + 2LH + LH + 1) per graph;
    and the level's time against the bound of that fused path on the matrix pipe (its intensity is far above the machine balance)."""
    ld = -(-N // 4) * 4
    plane_set = 4.0 * B * H * N * ld
    moved = sum(e["bytes_per_launch"] * e["launches"] for e in table.values()) / max(levels_timed, 1)
    alg = 4.0 * N * N * (2 + H + L * 2 * H + L * H + 1) * B
    F = fwd_flops_per_graph(N, H, L) * B
    if mode in BF16_PRODUCTS:
        bound_s = BF16_PRODUCTS[mode] * F / (BF16_MFMA_PEAK_TFLOPS * 1e12)
    else:
        bound_s = F / (F32_MFMA_PEAK_TFLOPS * 1e12)
    return {"plane_set_bytes": plane_set, "plane_sets_moved_per_level": moved / plane_set, "bytes_moved_per_level": moved,
            "algorithmic_bytes_per_level": alg, "moved_over_algorithmic": moved / alg,
            "level_ms": 1e3 * dt_level_s, "fused_matrix_bound_ms": 1e3 * bound_s, "frac_of_fused_bound": bound_s / dt_level_s,
            "hbm_bound_ms_at_moved_bytes": 1e3 * moved / (HBM_PEAK_GBS * 1e9),
            "note": "bytes are the library's per-launch algorithmic counts of the timed kernels (timed classes only: the diffusion kernels' "
                    "12 N^2 B bytes per level are not in it)"}


def dominant(table):
    """The kernel class with the largest summed device time in the timed region."""
    if not table:
        return None
    k = max(table, key=lambda n: table[n]["share_of_step_time"])
    e = dict(table[k])
    e["class"] = k
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nodes", type=int, default=200)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--levels", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32, help="graphs per GPU")
    ap.add_argument("--train-steps", type=int, default=None, help="timed optimizer steps for the training figure (default: max(10, --steps); 0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-f32", action="store_true", help="skip the exact-float32 comparison chains")
    ap.add_argument("--no-prof", action="store_true",
                    help="no per-kernel HIP events in the timed region (the roofline objects are then empty): for the launch-bound "
                         "small configurations, where two event records per kernel are a visible share of a 9 us launch")
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3", "bf16"],
                    help="matrix arithmetic of the sampler's H=128 kernels (include/dppm_hip.h DPPM_PREC_*)")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="rendezvous only (gloo, no GPU call): checks that --gpus N starts N ranks and relays rank 0's line")
    args = ap.parse_args()
    if args.train_steps is None:
        args.train_steps = max(10, args.steps)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: drop the torch.distributed environment (bench.py starts its own "
                         f"ranks) or launch with --nproc-per-node {args.gpus}")
    import torch.distributed as dist
    if args.selftest_launch:
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.tensor([rank + 1.0])
        if world > 1:
            dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"metric": "selftest", "value": float(t.item()), "n_gpus": world}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from discrete_dppm_graphs_amd import engine, ops
    from discrete_dppm_graphs_amd.model.ppgn import Powerful

    N, H, L, T, B = args.nodes, args.hidden, args.layers, args.levels, args.batch
    torch.manual_seed(1234)
    model = Powerful(num_layers=L, hidden=H, hidden_final=H, n_nodes=N, normalization="instance", adj_out=True,
                     input_features=2, dropout_p=1e-6).cuda().eval()
    model.precision = args.precision
    rng = np.random.default_rng(1234 + rank)
    flags = torch.ones(B, N, device="cuda")
    seed = 1234
    gid0 = rank * B

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], device="cuda", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    empty_prof = {}

    def timed_chains(steps, warmup):
        for w in range(warmup):
            engine.sample_chain_simple(model, flags, T, seed + w, gid0)
        barrier()
        if not args.no_prof:
            ops.prof_begin()
        t0 = time.perf_counter()
        for k in range(steps):
            out = engine.sample_chain_simple(model, flags, T, seed + 100 + k, gid0 + k * world * B)
        barrier()
        dt = time.perf_counter() - t0
        prof, truncated = ops.prof_end() if not args.no_prof else (empty_prof, False)
        assert out.shape == (B, N, N) and torch.equal(out, out.transpose(1, 2))
        return max_over_ranks(dt), prof, truncated

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")      # written by tools/pmc_aggregate.py from rocprofv3 --pmc passes
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("geometry") == {"nodes": N, "hidden": H, "layers": L, "graphs_per_gpu": B} and tj.get("precision", "bf16x3") == args.precision:
            traffic = tj.get("bytes_per_launch")

    # ------------------------------------------------------------------ sampling (headline)
    dt, prof, truncated = timed_chains(args.steps, args.warmup)
    graphs_per_s = args.steps * B * world / dt
    table = kernel_table(prof, dt, traffic, args.precision)
    path = path_object(table, N, H, L, B, args.steps * T, dt / (args.steps * T), args.precision) if table else None

    # the same chains with exact-float32 matrix products: same number of steps, its own kernel table
    exact = None
    if args.precision != "f32" and not args.no_exact_f32:
        model.precision = "f32"
        edt, eprof, _ = timed_chains(args.steps, 1)
        etable = kernel_table(eprof, edt, None, "f32")
        exact = {"value": args.steps * B * world / edt, "unit": "graphs/s", "ms_per_step": 1e3 * edt / args.steps, "steps": args.steps,
                 "dtype": "f32", "note": "same workload with DPPM_PREC_F32 (v_mfma_f32_16x16x4_f32) in every kernel",
                 "model_tflops": args.steps * B * world / edt * T * fwd_flops_per_graph(N, H, L) / 1e12,
                 "roofline": dominant(etable), "kernels": etable,
                 "path": path_object(etable, N, H, L, B, args.steps * T, edt / (args.steps * T), "f32") if etable else None}
        model.precision = args.precision

    # ------------------------------------------------------------------ training throughput (second half of the metric)
    train = None
    if args.train_steps > 0:
        model.train()
        state = engine.FlatState(model, lr=1e-3)
        adj = torch.as_tensor(sbm_graphs(rng, B, N)).cuda()
        sig = torch.as_tensor(np.random.default_rng(99).uniform(0.0, 0.5, size=B)).cuda()      # resident draws, as fit() keeps them
        loss = engine.train_step_simple(model, state, adj, sig, seed, 0, gid0, B_global=B * world)
        barrier()
        if not args.no_prof:
            ops.prof_begin()
        t0 = time.perf_counter()
        for k in range(args.train_steps):
            loss = engine.train_step_simple(model, state, adj, sig, seed, 1 + k, gid0, B_global=B * world)
        barrier()
        tdt = time.perf_counter() - t0
        tprof = ops.prof_end()[0] if not args.no_prof else empty_prof
        tdt = max_over_ranks(tdt)
        F = fwd_flops_per_graph(N, H, L)
        ttable = kernel_table(tprof, tdt, None, "train")
        train = {"train_steps_per_s": args.train_steps / tdt, "ms_per_train_step": 1e3 * tdt / args.train_steps,
                 "global_batch": B * world, "loss": float(loss.item()), "timed_steps": args.train_steps,
                 "dtype": "f32 (float32 tensors; forward linears exact-f32 MFMA, contraction / transposed linears / weight gradients f32x6 = three bf16 "
                          "pieces per operand on the bf16 pipe, float32-grade: DESIGN.md section 2)",
                 "model_tflops": 3 * F * B * world * args.train_steps / tdt / 1e12,
                 "roofline": dominant(ttable), "kernels": ttable}
        model.eval()

    if rank == 0:
        F = fwd_flops_per_graph(N, H, L)
        roofline = dominant(table)
        if roofline is not None:
            roofline["records_truncated"] = truncated
        cfg_names = {(200, 128, 8): "BASELINE configs[3] geometry", (512, 128, 8): "BASELINE configs[4] geometry",
                     (60, 128, 8): "BASELINE configs[2] geometry (sampling)", (20, 32, 6): "BASELINE configs[1] geometry",
                     (18, 32, 6): "BASELINE configs[0] geometry"}
        line = {
            "metric": f"graphs sampled/s (PPGN, N={N}); train steps/s in `train`", "value": graphs_per_s, "unit": "graphs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "f32 (tensors float32; H=128 products on bf16x3-split MFMA, float32 accumulate, 1e-3 parity bar held)",
                      "bf16": "bf16 products, float32 tensors and accumulation"}[args.precision],
            "data": "synthetic",
            "config": {"workload": f"ppgn_simple sampling on synthetic SBM N={N} ({cfg_names.get((N, H, L), 'custom geometry')}): {L}x{H} PPGN, "
                                   f"{T} noise levels, {B} graphs/GPU per chain", "nodes": N, "hidden": H, "layers": L,
                       "noise_levels": T, "graphs_per_gpu": B, "precision": args.precision,
                       "parallelism": f"graph-sharded x{world}, no collective"},
            "model_tflops": graphs_per_s * T * F / 1e12,
            "roofline": roofline,
            "path": path,
            "kernels": table,
            "exact_f32": exact,
            "train": train,
        }
        if not args.no_cpu_baseline and world == 1:     # the host-core baseline is reported at N = 1 only
            line["cpu_baseline"] = cpu_baseline(N, H, L, T)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
