#!/usr/bin/env python
"""Evaluation entry point, mirror of the reference's ``test.py`` (= its train1.py with ``--checkpoint`` instead of
``--resume`` and one ``validate()`` on the source and target test splits, test.py:157,192-226,584).

    python test.py data/H3D -t Hand3DStudio --checkpoint models/H3D_best_754.pth [--synthetic]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import torch

import mi355
import train1 as T
import uda.model as models
from uda.model.loss import JointsKLLoss
from uda.model.pose_resnet2 import Upsampling
from uda.model.regda_7 import PoseResNetx9
from utils.logger import CompleteLogger


def main(args):
    T.init_distributed()
    logger = CompleteLogger(args.log, 'test', quiet=T.RANK != 0)
    print(args)
    if T.device.type != 'cuda':
        raise SystemExit('this evaluation path needs an MI355X (HIP kernels only, no CPU fallback)')
    mi355.load()
    mi355.set_compute_dtype(args.dtype)
    _, val_s, _, val_t = T.build_datasets(args)
    val_source_loader, val_target_loader = T.make_loader(val_s, args, False), T.make_loader(val_t, args, False)
    backbone = models.__dict__[args.arch](pretrained=False)
    model = PoseResNetx9(backbone, Upsampling(backbone.out_features), 256, val_s.num_keypoints,
                         num_head_layers=args.num_head_layers, finetune=True).to(T.device)
    if args.checkpoint:
        ck = torch.load(args.checkpoint, map_location='cpu', weights_only=False)
        # the reference requires these keys (test.py:192-201); only `model` and `epoch` are needed to evaluate
        missing = [k for k in ('model',) if k not in ck]
        if missing:
            raise SystemExit('checkpoint lacks %s' % missing)
        model.load_state_dict(ck['model'])
        print('loaded checkpoint (epoch %s)' % ck.get('epoch'))
    criterion = JointsKLLoss()
    s_acc = T.validate(val_source_loader, model, criterion, args)
    t_acc = T.validate(val_target_loader, model, criterion, args)
    print("Source: {:4.3f} Target: {:4.3f}".format(s_acc['all'], t_acc['all']))
    for name, acc in t_acc.items():
        print("{}: {:4.3f}".format(name, acc))
    logger.close()


if __name__ == '__main__':
    p = T.build_parser('Evaluation for Keypoint Detection Domain Adaptation')
    p.add_argument('--checkpoint', type=str, default=None, help='where restore model parameters from.')
    main(p.parse_args())
