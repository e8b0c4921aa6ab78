"""Argument namespace with the reference's names and defaults (utils/args.py:3-89), table-driven."""
import argparse

_CAM = {'s': 0.1104, 'rho_1': 10.0, 'rho_2': 10.2, 'sigma_cam': 0.003, 'pixel_pitch': 5.86e-6}

# (flag, kwargs) per group; groups are selected by `mode` exactly as the reference does.
_BASIC = [
    ('--cuda', dict(type=str, default='cuda:0')),
    ('--model_path', dict(type=str, default='./pretrained_weights')),
    ('--img_size', dict(type=int, nargs=2, default=[147, 147])),
    ('--big_img_size', dict(type=int, nargs=2, default=[587, 587])),
    ('--R', dict(type=int, default=21)),
    ('--w', dict(type=float, default=1)),
    ('--alpha_lambda', dict(type=float, default=5e-3)),
    ('--cam_params', dict(type=dict, default=dict(_CAM))),
    ('--mag', dict(type=float, default=4)),
]
_TRIPLE = dict(type=float, nargs=3)
_GROUPS = {
    'data_gen_train_val': [
        ('--data_path', dict(type=str, default='./data/data_train_val')),
        ('--num_sample_train', dict(type=int, default=8000)),
        ('--num_sample_val', dict(type=int, default=2000)),
        ('--num_shape', dict(type=int, nargs=2, default=[15, 26])),
        ('--Z_range', dict(type=float, nargs=2, default=[0.75, 1.18])),
        ('--alpha', dict(type=float, nargs=2, default=[180.0, 200.0])),
        ('--sigma', dict(type=float, default=2)),
    ],
    'local_train': [
        ('--data_path', dict(type=str, default='./data/data_train_val/patches')),
        ('--log_path', dict(type=str, default='./logs')),
        ('--epoch_num', dict(type=int, default=1000)),
        ('--learning_rate', dict(type=float, default=6e-5)),
        ('--batch_size', dict(type=int, default=64)),
        ('--beta_bndry_loc', dict(type=float, default=0.001)),
        ('--beta_smthns', dict(type=float, default=0.0005)),
        ('--dynamic_epoch', dict(type=int, default=200)),
    ],
    'global_pre': [
        ('--stride', dict(type=int, default=2)),
        ('--data_path', dict(type=str, default='./data/data_train_val')),
        ('--batch_size', dict(type=int, default=1)),
    ],
    'global_train': [
        ('--stride', dict(type=int, default=2)),
        ('--data_path', dict(type=str, default='./data/data_train_val')),
        ('--log_path', dict(type=str, default='./logs')),
        ('--epoch_num', dict(type=int, default=350)),
        ('--learning_rate', dict(type=float, default=1e-4)),
        ('--batch_size', dict(type=int, default=8)),
        ('--gamma_color', dict(default=[1.0, 0.1, 0.1], **_TRIPLE)),
        ('--gamma_color_cons', dict(default=[0.2, 0.1, 0.05], **_TRIPLE)),
        ('--gamma_bndry_cons', dict(default=[0.05, 0.05, 0.02], **_TRIPLE)),
        ('--gamma_smthns', dict(default=[0.005, 0.1, 0.002], **_TRIPLE)),
        ('--gamma_smthns_cons', dict(default=[0.005, 0.1, 0.002], **_TRIPLE)),
        ('--gamma_bndry_loc', dict(default=[0.0001, 0.05, 0.0001], **_TRIPLE)),
        ('--gamma_depth', dict(default=[0.0001, 0.05, 0.5], **_TRIPLE)),
        ('--dynamic_epoch', dict(type=int, nargs=3, default=[30, 100, 200])),
        ('--input_size', dict(type=int, default=38)),
        ('--output_size', dict(type=int, default=12)),
        # not in the reference: continue a run from {model_path}/global_resume.ckpt (written after every epoch), and stop cleanly
        # after the epoch that passes this many seconds (0 = no limit) - a 350-epoch schedule in bounded jobs
        ('--resume', dict(action='store_true')),
        ('--time_budget', dict(type=float, default=0.0)),
    ],
    'data_gen_test': [
        ('--data_path', dict(type=str, default='./data/data_test')),
        ('--frgd_path', dict(type=str, default='./data/MS_COCO_annotations/')),
        ('--bkgd_path', dict(type=str, default='./data/Painting/')),
        ('--num_sample_test', dict(type=int, default=200)),
        ('--Z_range', dict(type=float, nargs=2, default=[0.75, 1.18])),
        ('--alpha', dict(type=int, nargs=2, default=[180, 200])),
        ('--sigma', dict(type=float, default=2)),
    ],
    'eval': [
        ('--stride', dict(type=int, default=2)),
        ('--log_path', dict(type=str, default='./logs')),
        ('--batch_size', dict(type=int, default=1)),
        ('--crop', dict(type=int, default=10)),
        ('--rho_prime', dict(type=float, default=10.39)),
        ('--densify', dict(type=str, default=None, choices=[None, 'pp', 'w'])),
    ],
}


def get_args(mode, big=False, argv=None):
    """Same call as the reference (`get_args('eval')`); `argv=[]` gives the defaults without reading sys.argv."""
    if mode not in _GROUPS:
        raise ValueError(f"unknown mode {mode!r}")
    parser = argparse.ArgumentParser()
    for flag, kw in _BASIC + _GROUPS[mode]:
        parser.add_argument(flag, **kw)
    if mode == 'eval':
        if big:
            parser.add_argument('--n_margin_patch', type=int, default=10)
            parser.add_argument('--data_path', type=str, default='./data/data_test_big')
        else:
            parser.add_argument('--data_path', type=str, default='./data/data_test')
    return parser.parse_args(argv)
