"""Mirror of the reference's utils/loading_utils.py for the PPGN path: model factory and checkpoint discovery."""
import logging
import os

import torch

from ..model.ppgn import Powerful
from ..model.ppgn_indnoise import Powerful_Ind, Powerful_sincos
from ..model.ppgn_indnoise_multdim import Powerful_Ind_Mult

# the noise-conditioning variants (reference :121-189): class, `name` keyword, input_features
_VARIANTS = {'ppgn_ind': (Powerful_Ind, "ppgn_indnoise", 2), 'ppgn_ind_mult': (Powerful_Ind_Mult, "ppgn_indnoise_mult", 2),
             'ppgn_ind_sincos': (Powerful_sincos, "ppgn_sincos", 1)}


def get_score_model(config, dev=None, **kwargs):
    """reference utils/loading_utils.py:62-195.  `ppgn` and its three noise-conditioning variants are built (the variants
    for inference / sampling, see model/ppgn_indnoise.py); the other model families the reference knows are rejected
    explicitly, unknown names raise ValueError like the reference (:193)."""
    if dev is None:
        dev = config.dev
    name = config.model.name
    if name == 'ppgn':
        score_model = Powerful(use_norm_layers=False,
                               name="ppgn",
                               channel_num_list=[],
                               feature_nums=[],
                               gnn_hidden_num_list=[],
                               num_layers=config.num_layers,
                               input_features=2,
                               hidden=config.hidden,
                               hidden_final=config.hidden_final,
                               dropout_p=config.dropout,
                               simplified=False,
                               n_nodes=config.n_nodes,
                               normalization=config.normalization,
                               cat_output=True,
                               adj_out=True,
                               output_features=1,
                               residual=False,
                               project_first=False,
                               node_out=False,
                               noise_mlp=config.noise_mlp,
                               config=config).to(dev)
    elif name in _VARIANTS:
        cls, ref_name, in_feat = _VARIANTS[name]
        score_model = cls(use_norm_layers=False, name=ref_name, channel_num_list=[], feature_nums=[], gnn_hidden_num_list=[],
                          num_layers=config.num_layers, input_features=in_feat, hidden=config.hidden,
                          hidden_final=config.hidden_final, dropout_p=config.dropout, simplified=False, n_nodes=config.n_nodes,
                          normalization=config.normalization, cat_output=True, adj_out=True, output_features=1, residual=False,
                          project_first=False, node_out=False, noise_mlp=config.noise_mlp, config=config).to(dev)
    elif name in ('gnn', 'mlp', 'cov', 'unet', 'edp-gnn'):
        raise NotImplementedError(f"model.name={name!r} is outside the accelerated hot path (PPGN family only)")
    else:
        raise ValueError(f'Unknown model name {name}')
    logging.info('model: ' + str(score_model))
    return score_model


def prepare_test_model_train(config, save_dir):
    """reference utils/loading_utils.py:223-234: every readable checkpoint in save_dir, largest sigma_list first.
    Checkpoints are dicts of tensors / python scalars written by `fit`; they are read with weights_only=True."""
    models = []
    config.model_files = os.listdir(save_dir)
    for file in config.model_files:
        path = os.path.join(save_dir, file)
        if os.path.isdir(path):
            continue
        try:
            ckp = torch.load(path, map_location=config.dev, weights_only=True)
            models.append((file, ckp['sigma_list'], [ckp, config.dev]))
        except Exception as e:   # the reference prints and skips unreadable files (:230-232)
            print(e)
            continue
    models.sort(key=lambda x: x[1], reverse=True)
    return models
