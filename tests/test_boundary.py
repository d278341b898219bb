"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/dppm_hip.h declares,
the product never imports the oracle, host-side mirrors behave like the reference's (names, errors, state_dict keys)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    declared = set()
    for h in ("dppm_hip.h", "dppm_hip_debug.h"):      # the drop-in interface, and the measurement / test hooks kept out of it
        header = open(os.path.join(ROOT, "include", h)).read()
        header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
        names = set(re.findall(r"\b(dppm_[a-z_A-Z0-9]+)\s*\(", header))
        assert (h == "dppm_hip.h") == (not any(n.startswith(("dppm_prof_", "dppm_test_", "dppm_debug_")) for n in names)), h
        declared |= names
    assert len(declared) >= 24
    from discrete_dppm_graphs_amd import _lib
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(_lib.lib, name) is not None
    assert _lib.lib.dppm_abi_version() == _lib.ABI_VERSION
    # size queries are pure host arithmetic and may run without a GPU
    assert _lib.lib.dppm_ppgn_workspace_bytes(2, 20, 32, 6, 0) > 0
    assert _lib.lib.dppm_ppgn_workspace_bytes(2, 20, 32, 6, 1) > _lib.lib.dppm_ppgn_workspace_bytes(2, 20, 32, 6, 0)
    assert _lib.lib.dppm_ppgn_workspace_bytes(0, 20, 32, 6, 0) == 0
    assert _lib.lib.dppm_loss_workspace_bytes(4, 10) >= 4 * 8


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "discrete_dppm_graphs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
    # and importing the product does not pull it in
    code = "import sys; import discrete_dppm_graphs_amd.ppgn_simple, discrete_dppm_graphs_amd.sample_ppgn_vlb; " \
           "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)


def test_ops_refuse_cpu_tensors_loudly():
    from discrete_dppm_graphs_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.edgeflip(torch.zeros(1, 4, 4), torch.ones(1, 4), 0.1)
    from discrete_dppm_graphs_amd.model.ppgn import Powerful
    m = Powerful(num_layers=1, hidden=8, hidden_final=8, normalization="instance", adj_out=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(None, torch.zeros(1, 4, 4), torch.ones(1, 4, 4), 0.1)


def test_powerful_state_dict_and_seeded_init_match_reference():
    """Same constructor order => torch.manual_seed(s) yields the reference's initial weights (fixture w:: entries)."""
    from discrete_dppm_graphs_amd.model.ppgn import Powerful
    g = np.load(os.path.join(G, "g1_ppgn_n6_h8_l2.npz"))
    torch.manual_seed(101)
    m = Powerful(use_norm_layers=False, name="ppgn", channel_num_list=[], feature_nums=[], gnn_hidden_num_list=[],
                 num_layers=2, input_features=2, hidden=8, hidden_final=8, dropout_p=1e-6, simplified=False, n_nodes=6,
                 normalization="instance", cat_output=True, adj_out=True, output_features=1, residual=False,
                 project_first=False, node_out=False, noise_mlp=False, config=None)
    sd = m.state_dict()
    ref_keys = sorted(k[3:] for k in g.files if k.startswith("w::"))
    assert sorted(sd.keys()) == ref_keys
    for k in ref_keys:
        assert np.array_equal(sd[k].numpy(), g["w::" + k]), k
    assert m.get_out_dim() == 1
    m.load_state_dict({k: torch.from_numpy(g["w::" + k]) for k in ref_keys}, strict=True)


def test_unsupported_configurations_raise():
    from discrete_dppm_graphs_amd.model.ppgn import Powerful
    with pytest.raises(ValueError):
        Powerful(normalization="bogus", adj_out=True)
    for kw in (dict(normalization="layer"), dict(residual=True), dict(adj_out=False), dict(node_out=True), dict(simplified=True)):
        args = dict(normalization="instance", adj_out=True)
        args.update(kw)
        with pytest.raises(NotImplementedError):
            Powerful(**args)
    from discrete_dppm_graphs_amd.utils.easydict import EasyDict
    from discrete_dppm_graphs_amd.utils.loading_utils import get_score_model
    cfg = EasyDict(dev="cpu", model=dict(name="nope"))
    with pytest.raises(ValueError, match="Unknown model name"):
        get_score_model(cfg)
    cfg.model.name = "edp-gnn"
    with pytest.raises(NotImplementedError):
        get_score_model(cfg)


def test_schedules_match_oracle_and_fixture():
    from discrete_dppm_graphs_amd import engine
    from oracle import diffusion as od
    g = np.load(os.path.join(G, "g5_sigma_lin.npz"))
    for T in (8, 16, 32, 64, 128):
        assert engine.sigma_lin(g[f"line_{T}"].tolist()) == g[f"beta_{T}"].tolist()
        assert engine.schedule_vlb(T) == od.schedule_vlb(T)
        s = od.schedule_simple(T)
        assert engine.schedule_simple(T) == [s[len(s) - c - 1] for c in range(T)]
    sched = engine.vlb_batch_schedule(16, [1, 16, 7])
    assert sched.shape == (3, 3) and sched[0, 2] == 0.0 and sched[1, 0] == 0.5


def test_level_tables_of_the_one_launch_chains_hold_the_per_level_roundings():
    """engine.level_table_simple / level_table_vlb (what dppm_ppgn_chain_* read per level): the network's noise input is the float32
    rounding model._prepare makes of the python double, the probabilities are formed in double and rounded once, as FlipOp /
    make_vlb_consts of csrc/dppm_diffusion.hip do -- that is what makes the one-launch chain bit-identical to the per-level calls."""
    from discrete_dppm_graphs_amd import engine
    for T in (4, 16, 64):
        s = engine.schedule_simple(T)
        tab = engine.level_table_simple(s, "cpu").numpy()
        assert tab.shape == (T, 3) and tab.dtype == np.float32
        for c, sg in enumerate(s):
            assert tab[c, 0] == np.float32(sg) and tab[c, 1] == np.float32(1.0 - sg) and tab[c, 2] == np.float32(sg)
        v = engine.schedule_vlb(T)
        tv = engine.level_table_vlb(v, "cpu").numpy()
        assert tv.shape == (T, 7) and tv.dtype == np.float32
        for c, (st, bt, st1) in enumerate(v):
            want = [st, 1.0 - bt, bt, st1, 1.0 - st1, st, 1.0 - st]
            assert tv[c].tolist() == [float(np.float32(x)) for x in want]
    assert engine.level_table_simple(engine.schedule_simple(16), "cpu") is engine.level_table_simple(engine.schedule_simple(16), "cpu")   # cached


def test_configs_parse_and_carry_reference_keys(tmp_path, monkeypatch):
    from discrete_dppm_graphs_amd.utils import arg_helper
    monkeypatch.chdir(tmp_path)
    for name in ("train_ego_small_ddpm", "train_com_small_ddpm", "train_plan_ddpm", "train_sbm_n200_ddpm"):
        args = arg_helper.parse_arguments(argv=["-c", os.path.join(ROOT, "config", name + ".yaml"), "-m", "unit test"])
        cfg = arg_helper.get_config(args)
        for key in ("num_layers", "hidden", "hidden_final", "dropout", "n_nodes", "normalization", "noise_mlp", "num_levels",
                    "noisetype", "samplesize", "eval_from", "finalinterval", "seed"):
            assert key in cfg, (name, key)
        assert cfg.dataset.max_node_num > 0 and cfg.train.lr_dacey == 0.999 and cfg.test.split == 0.2
        assert os.path.isfile(os.path.join(cfg.save_dir, "config.yaml")) and os.path.isdir(cfg.model_save_dir)
        assert arg_helper.edict2dict(cfg)["dataset"]["name"] == cfg.dataset.name
    with pytest.raises(SystemExit):
        arg_helper.parse_arguments(argv=[])     # -c is required, as in the reference


def test_graphs_to_tensor_pads_and_flags():
    import networkx as nx
    from discrete_dppm_graphs_amd.utils.arg_helper import graphs_to_tensor
    from discrete_dppm_graphs_amd.utils.easydict import EasyDict
    from discrete_dppm_graphs_amd.utils.graph_utils import mask_adjs, pad_adjs
    cfg = EasyDict(dataset=dict(max_node_num=7))
    adjs, x = graphs_to_tensor(cfg, [nx.path_graph(4), nx.cycle_graph(7), np.ones((3, 3)) - np.eye(3)])
    assert adjs.shape == (3, 7, 7) and x.shape == (3, 7, 1) and cfg.dataset.in_feature == 1
    assert adjs[0].sum() == 6 and adjs[0, 4:].sum() == 0 and adjs[2].sum() == 6
    flags = adjs.sum(-1).gt(1e-5).float()
    assert flags.sum(-1).tolist() == [4, 7, 3]
    assert torch.equal(mask_adjs(torch.ones(3, 7, 7), flags), flags[:, :, None] * flags[:, None, :])
    with pytest.raises(ValueError):
        pad_adjs(np.zeros((5, 5)), 3)


def test_degree_mmd_hook():
    from discrete_dppm_graphs_amd.evaluation import stats
    a = [np.ones((4, 4)) - np.eye(4)] * 3
    assert stats.eval_graph_list(a, a)["degree"] == pytest.approx(0.0, abs=1e-12)
    stats.set_evaluator(lambda r, p, methods=None: {"degree": 1.0, "cluster": 2.0})
    try:
        assert stats.eval_graph_list(a, a) == {"degree": 1.0, "cluster": 2.0}
    finally:
        stats.set_evaluator(None)
    gs = stats.adjs_to_graphs(np.stack(a))
    assert gs[0].number_of_nodes() == 4 and gs[0].number_of_edges() == 6
