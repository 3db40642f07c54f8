"""Default hyper-parameters of the D3Feat network and loss, as plain attributes.

Values are the reference's argparse defaults (config.py:29-59,78-86) and the architecture list built in
training_3DMatch.py:44-56; blocks read them by attribute (blocks.py:557-578), so any namespace works.
"""
import types


def d3feat_architecture(num_layers=5):
    arch = ['simple', 'resnetb']
    for _ in range(num_layers - 1):
        arch += ['resnetb_strided', 'resnetb', 'resnetb']
    for _ in range(num_layers - 2):
        arch += ['nearest_upsample', 'unary']
    arch += ['nearest_upsample', 'last_unary']
    return arch


def default_config(**overrides):
    cfg = types.SimpleNamespace(
        num_layers=5, in_points_dim=3, first_features_dim=128, first_subsampling_dl=0.03, in_features_dim=1,
        conv_radius=2.5, deform_radius=5.0, num_kernel_points=15, KP_extent=2.0, KP_influence='linear',
        aggregation_mode='sum', fixed_kernel_points='center', use_batch_norm=False, batch_norm_momentum=0.02,
        deformable=False, modulated=False,
        dist_type='euclidean', desc_loss='circle', pos_margin=0.1, neg_margin=1.4, m=0.1, log_scale=10,
        safe_radius=0.1, det_loss='score', desc_loss_weight=1.0, det_loss_weight=1.0,
        optimizer='SGD', lr=0.01, weight_decay=1e-6, momentum=0.98, scheduler='ExpLR', scheduler_gamma=0.1 ** (1 / 80),
        num_node=128, downsample=0.03, batch_size=1)
    cfg.architecture = d3feat_architecture(cfg.num_layers)
    for k, v in overrides.items():
        setattr(cfg, k, v)
    if 'num_layers' in overrides and 'architecture' not in overrides:
        cfg.architecture = d3feat_architecture(cfg.num_layers)
    return cfg
