"""Device-resident dataset: after the first step of a pass no training step, and after the
index table no sampling batch, copies anything between host and device."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _toy(tmp_path, monkeypatch, which):
    from discrete_dppm_graphs_amd.utils import arg_helper
    from discrete_dppm_graphs_amd.utils.easydict import EasyDict
    from test_gpu_end_to_end import rand_graphs
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(21)
    os.makedirs("data")
    adjs = rand_graphs(rng, 30, 10)
    np.savez(os.path.join("data", "toy.npz"), adjs=adjs, sizes=(adjs.sum(-1) > 0).sum(-1))
    cfg = EasyDict(exp_dir="exp", exp_name="toy", seed=1234, dataset=dict(dataset_size=30, max_node_num=10, name="toy"),
                   model=dict(name="ppgn"), test=dict(batch_size=4, split=0.2),
                   train=dict(batch_size=5, lr_dacey=0.999, lr_init=0.01, max_epoch=3, sample_interval=100, save_interval=1, weight_decay=0.0),
                   eval_from=1000, samplesize=8, finalinterval=1000, num_layers=2, hidden=16, hidden_final=16, dropout=1e-6, n_nodes=10,
                   normalization="instance", noise_mlp=False, num_levels=[4], noisetype="switched", assert_no_host_sync=True)
    arg_helper.process_config(cfg)
    return cfg


@pytest.mark.parametrize("which", ["simple", "vlb"])
def test_training_steps_make_no_host_transfer(tmp_path, monkeypatch, which):
    """train_main over a ResidentLoader with config.assert_no_host_sync: from the second step of every pass on,
    torch.cuda.set_sync_debug_mode("error") is active, so any synchronising host<->device copy (a batch upload, a per-step
    sigma upload, an .item()) raises.  Host-side uploads are counted as well: one permutation per training / test pass and one
    draw table per pass."""
    import discrete_dppm_graphs_amd.ppgn_simple as ps
    import discrete_dppm_graphs_amd.ppgn_vlb as pv
    from discrete_dppm_graphs_amd.utils import resident
    cfg = _toy(tmp_path, monkeypatch, which)
    n0 = resident.upload.calls
    state = (ps if which == "simple" else pv).train_main(cfg, types.SimpleNamespace(log_level="INFO", comment="r", resume=""))
    assert state.step_count == 3 * 5                       # 24 training graphs / batch 5 -> 5 steps per epoch (tail of 4)
    # per epoch: train permutation + train draws + test permutation + test draws
    assert resident.upload.calls - n0 == 3 * 4
    assert torch.cuda.get_sync_debug_mode() == 0           # restored


def test_sampling_pool_is_indexed_on_the_device(tmp_path, monkeypatch):
    from discrete_dppm_graphs_amd import engine, sample_ppgn_simple as sp
    from discrete_dppm_graphs_amd.utils import resident
    from discrete_dppm_graphs_amd.utils.loading_utils import get_score_model
    cfg = _toy(tmp_path, monkeypatch, "simple")
    model = get_score_model(cfg).eval()
    pool = torch.as_tensor(np.load(os.path.join("data", "toy.npz"))["adjs"]).cuda()
    calls = []

    def chain(model, flags, noise_num, seed, graph_id0=0):
        assert flags.is_cuda and flags.shape == (4, 10)
        calls.append(graph_id0)
        if len(calls) > 1:      # every batch after the first: nothing may cross the host boundary until the chain is done
            torch.cuda.set_sync_debug_mode("error")
        try:
            return engine.sample_chain_simple(model, flags, noise_num, seed, graph_id0)
        finally:
            torch.cuda.set_sync_debug_mode(0)

    import networkx as nx
    ref = [nx.complete_graph(4), nx.cycle_graph(5), nx.path_graph(6), nx.star_graph(4)]
    n0 = resident.upload.calls
    res, graphs = sp.run_sample(cfg, model, pool, 12, 4, chain, ref)
    assert calls == [0, 4, 8] and len(graphs) == 12
    assert resident.upload.calls - n0 == 1                 # the pool-index table of the whole call
    # the same call again draws a different table (keyed by the call number), the same call number the same table
    sp._CALLS["n"] -= 1
    res2, graphs2 = sp.run_sample(cfg, model, pool, 12, 4, engine.sample_chain_simple, ref)
    assert all(nx.utils.graphs_equal(a, b) for a, b in zip(graphs, graphs2))
