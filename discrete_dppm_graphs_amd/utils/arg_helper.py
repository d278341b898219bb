"""Mirror of the reference's utils/arg_helper.py: CLI, YAML config, seeding/logging, data -> tensors, load_model."""
import argparse
import logging
import os
import pickle
import random
import sys
import time

import numpy as np
import torch
import yaml

from .easydict import EasyDict as edict
from .graph_utils import pad_adjs, set_rng
from .loading_utils import get_score_model
from .resident import ResidentLoader


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def parse_arguments(default_config="train.yaml", argv=None):
    """reference utils/arg_helper.py:22-40 (same three flags)."""
    parser = argparse.ArgumentParser(description="Running Experiments")
    parser.add_argument('-c', '--config_file', type=str, default=os.path.join('config', default_config), required=True,
                        help="Path of config file")
    parser.add_argument('-l', '--log_level', type=str, default='INFO',
                        help="Logging Level, one of: DEBUG, INFO, WARNING, ERROR, CRITICAL")
    parser.add_argument('-m', '--comment', type=str, default="", help="A single line comment for the experiment")
    parser.add_argument('--resume', type=str, default="", help="checkpoint (.pth written by this package) to continue training from")
    return parser.parse_args(argv)


def get_config(args):
    """reference :43-47.  SafeLoader: configs are plain data."""
    with open(args.config_file, 'r') as f:
        config = edict(yaml.load(f, Loader=yaml.SafeLoader))
    process_config(config, comment=args.comment)
    return config


def process_config(config, comment=''):
    """reference :50-76.  Under torch.distributed (one process per GPU) the run directory is named by rank 0 and handed to
    the other ranks, so that every rank reads the checkpoints rank 0 writes; only rank 0 creates directories and the
    config snapshot."""
    dist, rank, world = _dist()
    if not torch.cuda.is_available():
        config.dev = torch.device("cpu")
    else:       # one process per GPU: the rank's own device (set by the launcher code before the process group exists)
        config.dev = torch.device("cuda", torch.cuda.current_device()) if world > 1 else torch.device("cuda")
    names = [str(os.getpid()), time.strftime('%b-%d-%H-%M-%S')]
    if world > 1:
        dist.broadcast_object_list(names, src=0)
    config.run_id = names[0]
    config.folder_name = '_'.join([config.model.name, config.dataset.name, comment.replace(' ', '_'), names[1], config.run_id])
    if 'save_dir' not in config:
        config.save_dir = os.path.join(config.exp_dir, config.exp_name, config.folder_name)
    if 'model_save_dir' not in config:
        config.model_save_dir = os.path.join(config.save_dir, 'models')
    if rank == 0:
        mkdir(config.exp_dir)
        mkdir(config.save_dir)
        mkdir(config.model_save_dir)
        save_name = os.path.join(config.save_dir, 'config.yaml')
        with open(save_name, 'w') as f:
            yaml.dump(edict2dict(config), f, default_flow_style=False)
    if world > 1:
        dist.barrier()


def edict2dict(edict_obj):
    """reference :80-89, with values made YAML/weights_only friendly (devices become strings)."""
    dict_obj = {}
    for key, vals in edict_obj.items():
        if isinstance(vals, dict):
            dict_obj[key] = edict2dict(vals)
        elif isinstance(vals, torch.device):
            dict_obj[key] = str(vals)
        else:
            dict_obj[key] = vals
    return dict_obj


def mkdir(folder):
    if not os.path.isdir(folder):
        os.makedirs(folder)


def set_seed(config, args=None):
    """reference :116-127 (plus the Philox stream of the HIP kernels).  The reference's `np.random.RandomState(seed)`
    (:101) builds and drops a generator, leaving numpy's global state unseeded; nothing on this package's path draws from
    that global state (utils/resident.host_rng), it is seeded here only for user code."""
    random.seed(config.seed)
    torch.manual_seed(config.seed)
    torch.cuda.manual_seed_all(config.seed)
    np.random.seed(int(config.seed) & 0xFFFFFFFF)
    set_rng(config.seed)


def set_seed_and_logger(config, args):
    """reference :97-114."""
    set_seed(config)
    _, rank, _ = _dist()
    log_file = os.path.join(config.save_dir, args.log_level.lower() + (".log" if rank == 0 else f".rank{rank}.log"))
    FORMAT = args.comment + '| %(asctime)s %(message)s'
    fh = logging.FileHandler(log_file)
    fh.setLevel(args.log_level)
    logging.basicConfig(level=logging.DEBUG, format=FORMAT, datefmt='%m-%d %H:%M:%S',
                        handlers=[fh, logging.StreamHandler(sys.stdout)])
    logging.info('EXPERIMENT BEGIN: ' + args.comment)
    logging.info('logging into %s', log_file)


def graphs_to_tensor(config, graph_list):
    """reference :137-194: networkx graphs (or dense numpy adjacencies) -> padded [G, N, N] adjacency and [G, N, F]
    standardised node features.  nx.to_numpy_matrix (:175) no longer exists in networkx 3; to_numpy_array is used."""
    import networkx as nx
    adjs_list, x_list = [], []
    feature_len = None
    N = config.dataset.max_node_num
    for g in graph_list:
        if isinstance(g, np.ndarray):
            adj = np.asarray(g, dtype=np.float64)
            feature_np = np.ones([adj.shape[0], 1])
        else:
            assert isinstance(g, nx.Graph)
            feature_list, node_list = [], []
            for v, feature in g.nodes.data('feature'):
                node_list.append(v)
                if feature is None:
                    feature = [1.0]
                elif isinstance(feature, (float, int)):
                    feature = [feature]
                else:
                    feature = list(feature)
                feature_list.append(feature)
            feature_np = np.asarray(feature_list, dtype=np.float64)
            adj = nx.to_numpy_array(g, nodelist=node_list)
        if feature_len is None:
            feature_len = feature_np.shape[-1]
        else:
            assert feature_len == feature_np.shape[-1]
        feature_np = feature_np - np.mean(feature_np, axis=0)
        feature_np = feature_np / (np.std(feature_np, axis=0) + 1e-6)
        x_list.append(np.concatenate([feature_np, np.zeros([N - feature_np.shape[0], feature_len])], axis=0))
        adjs_list.append(pad_adjs(adj, node_number=N))
    config.dataset.in_feature = feature_len
    adjs_tensor = torch.tensor(np.asarray(adjs_list), dtype=torch.float32)
    x_tensor = torch.tensor(np.asarray(x_list), dtype=torch.float32)
    return adjs_tensor, x_tensor


def graphs_to_dataloader(config, graph_list):
    """reference :130-134 (DataLoader(TensorDataset(adjs, x), batch_size, shuffle=True)), with the padded tensors uploaded
    once and batches indexed on the device (utils/resident.py)."""
    adjs_tensor, x_tensor = graphs_to_tensor(config, graph_list)
    return ResidentLoader(adjs_tensor, x_tensor, batch_size=config.train.batch_size, shuffle=True,
                          seed=int(getattr(config, "seed", 0)), device=config.dev)


def load_dataset(data_dir='data', file_name=None):
    """reference utils/data_generators.py:133-147.  Accepts `<name>.npz` (key `adjs`: object array or [G,N,N] dense
    adjacencies; preferred, nothing is executed on load) or the reference's `<name>.pkl` list of networkx graphs
    (unpickling runs code from the file: only do that with files you trust)."""
    base = os.path.join(data_dir, file_name)
    if os.path.exists(base + '.npz'):
        with np.load(base + '.npz', allow_pickle=False) as z:
            adjs = z['adjs']
            sizes = z['sizes'] if 'sizes' in z.files else [a.shape[0] for a in adjs]
        return [np.asarray(a[:n, :n]) for a, n in zip(adjs, sizes)], set()
    with open(base + '.pkl', 'rb') as f:
        graph_list = pickle.load(f)
    return graph_list, set()


def load_data(config, get_graph_list=False):
    """reference :197-205."""
    data_dir = config.dataset.get('data_dir', 'data') if isinstance(config.dataset, dict) else 'data'
    graph_list, _ = load_dataset(data_dir=data_dir, file_name=config.dataset.name)
    graph_list = graph_list[:config.dataset.dataset_size]
    test_size = int(config.test.split * len(graph_list))
    train_graph_list, test_graph_list = graph_list[test_size:], graph_list[:test_size]
    if get_graph_list:
        return train_graph_list, test_graph_list
    return graphs_to_dataloader(config, train_graph_list), graphs_to_dataloader(config, test_graph_list)


def load_model(ckp, dev):
    """reference :237-243."""
    model_config = edict(ckp['config'])
    model = get_score_model(config=model_config, dev=dev)
    model.load_state_dict(ckp['model'], strict=False)
    model.to(dev)
    model.eval()
    return model
