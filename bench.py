#!/usr/bin/env python3
"""Benchmark of the discrete graph-DDPM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[3], the configuration the metric is quoted on): synthetic SBM graphs with N = 200 nodes,
the 8 x 128 PPGN of gridsearch_ppgn_simple.py:19, `ppgn_simple` sampling with 64 noise levels, 32 graphs per GPU.
One step = one full reverse chain (x_T init, then 64 x [re-noise, denoiser forward, reverse step]) over one batch; graphs
are independent, so ranks take disjoint graph ids and exchange nothing (weak scaling).  `value` = graphs finished per
second over all ranks.  The sampler's denoiser runs in `--precision` (default bf16x3: float32 tensors, H=128 matrix
products as three bf16 MFMAs on split operands with float32 accumulation, held to the float32 parity bar by
tests/test_gpu_ppgn.py); `exact_f32` carries the same chain with exact-float32 MFMAs for comparison.  The same line also carries the training throughput (optimizer steps/s at global batch
32 x N GPUs, data parallel with one flat-gradient all-reduce per step), the roofline of the dominant kernel measured
live with HIP events, and a CPU baseline (the oracle's torch-CPU restatement on the host cores, bounded sample).
"""
import argparse
import json
import os
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


def cpu_baseline(N, H, L, T, seconds_budget=20.0):
    """The oracle (oracle/ppgn.py + oracle/diffusion.py, torch CPU fp32) on a bounded sample of the same workload:
    2 graphs through as many chain steps as fit the budget (>= 2), extrapolated to the full T-step chain."""
    from oracle import diffusion as od, ppgn as op
    rng = np.random.default_rng(1234)
    Bc = 2
    A = sbm_graphs(rng, Bc, N)
    flags = (A.sum(-1) > 1e-5).astype(np.float32)
    mask = od.generate_mask(flags)
    W = op.init_weights(L, H, seed=1234)
    x = od.gen_init_data(flags, rng.random((Bc, N, N), dtype=np.float32))
    sig = od.schedule_simple(T)
    steps, t0 = 0, time.perf_counter()
    with torch.no_grad():
        while steps < 2 or (time.perf_counter() - t0 < seconds_budget and steps < T):
            s = sig[len(sig) - steps - 1]
            x, _ = od.take_step_simple(x, lambda a: op.powerful_forward(W, a, mask, s, L).squeeze(-1), flags, s,
                                       rng.random((Bc, N, N), dtype=np.float32), rng.random((Bc, N, N), dtype=np.float32))
            steps += 1
    dt = time.perf_counter() - t0
    per_graph_chain = dt / steps * T / Bc
    return {"value": 1.0 / per_graph_chain, "unit": "graphs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{Bc} graphs x {steps} of {T} chain steps (N={N}, {L}x{H} PPGN, torch CPU fp32 oracle), extrapolated to the full chain"}


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
    ap.add_argument("--train-steps", type=int, default=3, help="timed optimizer steps for the training figure (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-f32", action="store_true", help="skip the exact-float32 comparison chain")
    ap.add_argument("--no-prof", action="store_true",
                    help="no per-kernel HIP events in the timed region (the roofline objects are then empty): for the launch-bound "
                         "small configurations, where two event records per kernel are a visible share of a 9 us launch")
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3", "bf16"],
                    help="matrix arithmetic of the sampler's H=128 kernels (include/dppm_hip.h DPPM_PREC_*)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    import torch.distributed as dist
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

    # ------------------------------------------------------------------ sampling (headline)
    for w in range(args.warmup):
        engine.sample_chain_simple(model, flags, T, seed + w, gid0)
    barrier()
    if not args.no_prof:
        ops.prof_begin()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = engine.sample_chain_simple(model, flags, T, seed + 100 + k, gid0 + k * world * B)
    barrier()
    dt = time.perf_counter() - t0
    prof, truncated = ops.prof_end() if not args.no_prof else ({k: {"ms": 0.0, "work": 0.0, "launches": 0} for k in ops.PROF_CLASSES}, False)
    tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    assert out.shape == (B, N, N) and torch.equal(out, out.transpose(1, 2))
    graphs_per_s = args.steps * B * world / dt

    # the same chain with exact-float32 matrix products (one warm-up chain, one timed), rank 0's clock
    exact = None
    if args.precision != "f32" and not args.no_exact_f32:
        model.precision = "f32"
        engine.sample_chain_simple(model, flags, T, seed, gid0)
        barrier()
        t0 = time.perf_counter()
        engine.sample_chain_simple(model, flags, T, seed + 100, gid0)
        barrier()
        edt = time.perf_counter() - t0
        exact = {"value": B * world / edt, "unit": "graphs/s", "ms_per_step": 1e3 * edt, "steps": 1,
                 "note": "same workload with DPPM_PREC_F32 (v_mfma_f32_16x16x4_f32) in every kernel"}
        model.precision = args.precision

    # ------------------------------------------------------------------ training throughput (second half of the metric)
    train = None
    if args.train_steps > 0:
        model.train()
        state = engine.FlatState(model, lr=1e-3)
        adj = torch.as_tensor(sbm_graphs(rng, B, N)).cuda()
        sig = lambda: np.random.default_rng(99).uniform(0.0, 0.5, size=B)
        loss = engine.train_step_simple(model, state, adj, sig(), seed, 0, gid0, B_global=B * world)
        barrier()
        if not args.no_prof:
            ops.prof_begin()
        t0 = time.perf_counter()
        for k in range(args.train_steps):
            loss = engine.train_step_simple(model, state, adj, sig(), seed, 1 + k, gid0, B_global=B * world)
        barrier()
        tdt = time.perf_counter() - t0
        tprof = ops.prof_end()[0] if not args.no_prof else {}
        tm = torch.tensor([tdt], device="cuda", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        tdt = float(tm.item())
        F = fwd_flops_per_graph(N, H, L)
        train = {"train_steps_per_s": args.train_steps / tdt, "ms_per_train_step": 1e3 * tdt / args.train_steps,
                 "global_batch": B * world, "loss": float(loss.item()),
                 "model_tflops": 3 * F * B * world * args.train_steps / tdt / 1e12,
                 "kernel_classes": {k: {"ms_per_step": v["ms"] / args.train_steps,
                                        "tflops": (v["work"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and k != "inorm" else None}
                                    for k, v in tprof.items()}}
        model.eval()

    if rank == 0:
        pmm, nrm = prof["plane_matmul"], prof["inorm"]
        lin_keys = ("linear_gemm", "wstat_mlp2", "wstat_sum2", "wstat_cols")
        lin = {f: sum(prof[k][f] for k in lin_keys) for f in ("ms", "work", "launches")}
        tf = lambda c: (c["work"] / (c["ms"] * 1e-3) / 1e12) if c["ms"] > 0 else 0.0
        # dominant kernel: the fused MLP pair when H = 128 (weight-stationary path), else the generic linear GEMM
        dom_key = "wstat_mlp2" if prof["wstat_mlp2"]["launches"] > 0 else "linear_gemm"
        dom = prof[dom_key]
        split = args.precision != "f32" and dom_key == "wstat_mlp2"
        dom_name = {"wstat_mlp2": "ws_mlp_kernel<false,false,false> (both per-position MLPs of a PPGN layer, v_mfma_f32_16x16x4_f32; "
                                  "8*H^2*P*B FLOP per launch)",
                    "linear_gemm": "gemm_f32_kernel<...,TAG=0> (per-position linear layers, v_mfma_f32_32x32x2_f32)"}[dom_key]
        if split:
            dom_name = ("w3_mlp_kernel<%d,%d> (both per-position MLPs of a PPGN layer; float32 in HBM, operands split into bf16 pieces in "
                        "LDS, v_mfma_f32_16x16x32_bf16; algorithmic traffic 3*B*H*P*4 bytes per launch: read X, write o1 and o2)"
                        % ((3, 1) if args.precision == "bf16x3" else (1, 2)))
        F = fwd_flops_per_graph(N, H, L)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")      # written by tools/pmc_aggregate.py from rocprofv3 --pmc passes
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("geometry") == {"nodes": N, "hidden": H, "layers": L, "graphs_per_gpu": B}:
                traffic = tj.get("bytes_per_launch", {}).get(dom_key)
        avg_ms = dom["ms"] / max(dom["launches"], 1)
        common = {"traffic": traffic, "launches": dom["launches"], "avg_launch_ms": avg_ms,
                  "flops_per_launch": dom["work"] / max(dom["launches"], 1), "share_of_step_time": dom["ms"] * 1e-3 / dt,
                  "records_truncated": truncated}
        if split:     # HBM-bound: 3 plane sets of B*H*P floats per launch (P = N * round_up(N, 4))
            alg_bytes = 3.0 * B * H * N * ((N + 3) // 4 * 4) * 4
            gbs = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            nterm = 3 if args.precision == "bf16x3" else 1
            roofline = {"kernel": dom_name, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "bytes_per_launch": alg_bytes, "f32_equivalent_tflops": tf(dom),
                        "bf16_mfma_frac": nterm * tf(dom) / BF16_MFMA_PEAK_TFLOPS, **common}
        else:
            roofline = {"kernel": dom_name, "bound": "mfma", "achieved": tf(dom), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tf(dom) / F32_MFMA_PEAK_TFLOPS, **common}
        mm_hbm = None
        if split and pmm["ms"] > 0:
            mm_bytes = 3.0 * B * H * N * ((N + 3) // 4 * 4) * 4 * pmm["launches"]
            mm_hbm = {"achieved": mm_bytes / (pmm["ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": mm_bytes / (pmm["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        line = {
            "metric": "graphs sampled/s (PPGN, N=200); train steps/s in `train`", "value": graphs_per_s, "unit": "graphs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "f32 (tensors float32; H=128 products on bf16x3-split MFMA, float32 accumulate, 1e-3 parity bar held)",
                      "bf16": "bf16 products, float32 tensors and accumulation"}[args.precision],
            "data": "synthetic",
            "config": {"workload": f"ppgn_simple sampling on synthetic SBM N={N} (BASELINE configs[3]): {L}x{H} PPGN, "
                                   f"{T} noise levels, {B} graphs/GPU per chain", "nodes": N, "hidden": H, "layers": L,
                       "noise_levels": T, "graphs_per_gpu": B, "parallelism": f"graph-sharded x{world}, no collective"},
            "model_tflops": graphs_per_s * T * F / 1e12,
            "roofline": roofline,
            "roofline_linear_all": {"kernel": "every per-position linear layer (w3_mlp_kernel, w3_lin2_kernel, w3_lin_kernel in the split modes; "
                                              "ws_mlp_kernel, ws_lin_kernel<2>, ws_lin_kernel<1> in f32; generic GEMM when H != 128)",
                                    "bound": "hbm (f32-equivalent TFLOP/s shown against the f32 MFMA peak it would need)" if split else "mfma",
                                    "achieved": tf(lin), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": tf(lin) / F32_MFMA_PEAK_TFLOPS, "share_of_step_time": lin["ms"] * 1e-3 / dt,
                                    "by_kernel": {k: {"tflops": tf(prof[k]), "launches": prof[k]["launches"],
                                                      "avg_launch_ms": prof[k]["ms"] / max(prof[k]["launches"], 1)} for k in lin_keys}},
            "roofline_matmul": {"kernel": ("plane3_kernel (per-channel N x N contraction, split-bf16 operands on v_mfma_f32_16x16x32_bf16, model/ppgn.py:62)"
                                           if split else
                                           "plane_mm16_kernel / plane_mm_kernel (per-channel N x N contraction on v_mfma_f32_16x16x4_f32, model/ppgn.py:62)"),
                                "bound": "mfma", "achieved": tf(pmm), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": tf(pmm) / F32_MFMA_PEAK_TFLOPS, "hbm": mm_hbm,
                                "launches": pmm["launches"], "avg_launch_ms": pmm["ms"] / max(pmm["launches"], 1),
                                "share_of_step_time": pmm["ms"] * 1e-3 / dt},
            "roofline_inorm": {"bound": "hbm", "achieved": (nrm["work"] / (nrm["ms"] * 1e-3) / 1e9) if nrm["ms"] > 0 else 0.0,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": (nrm["work"] / (nrm["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if nrm["ms"] > 0 else 0.0,
                               "avg_launch_ms": nrm["ms"] / max(nrm["launches"], 1),
                               "share_of_step_time": nrm["ms"] * 1e-3 / dt},
            "exact_f32": exact,
            "train": train,
        }
        if not args.no_cpu_baseline and world == 1:     # the host-core baseline is reported at N = 1 only
            line["cpu_baseline"] = cpu_baseline(N, H, L, T)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
