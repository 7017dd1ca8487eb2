"""train1.py / test.py command lines: every flag of the reference's parser exists with the reference's default
(train1.py:602-674); on the GPU a tiny synthetic run writes the reference's checkpoint layout, resumes and
evaluates."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import PKG

REF_DEFAULTS = dict(source_root='data/RHD', source='RenderedHandPose', target=None, resize_scale=(0.6, 1.3), rotation=180,
                    image_size=256, heatmap_size=64, arch='resnet101', arch2='net_hg', pretrain='models/pretrain_rhd.pth',
                    ema_model=None, resume=None, resume2=None, num_head_layers=2, margin=4., trade_off=1., batch_size=32,
                    lr=0.01, momentum=0.9, wd=0.0001, lr_gamma=0.0001, lr_decay=0.75, lr_step=[45, 60], lr_factor=0.1,
                    workers=4, pretrain_epochs=70, epochs=200, iters_per_epoch=500, print_freq=100, seed=1, log='logs/mt',
                    phase='train', debug=False, ema_decay=0.999)


def test_parser_matches_reference_flags_and_defaults():
    import train1
    a = train1.build_parser().parse_args(['data/H3D'])
    assert a.target_root == 'data/H3D'
    for k, v in REF_DEFAULTS.items():
        assert getattr(a, k) == v, k
    b = train1.build_parser().parse_args(['d', '-s', 'X', '-t', 'Hand3DStudio', '-a', 'resnet50', '-b', '64', '--lr', '0.02', '-j', '2',
                                          '-i', '10', '-p', '5', '--wd', '1e-3', '--phase', 'test', '--debug', '--synthetic'])
    assert (b.source, b.target, b.arch, b.batch_size, b.lr, b.workers, b.iters_per_epoch, b.print_freq, b.wd, b.phase, b.debug,
            b.synthetic) == ('X', 'Hand3DStudio', 'resnet50', 64, 0.02, 2, 10, 5, 1e-3, 'test', True, True)


def test_synthetic_dataset_contract():
    from utils.synthetic_dataset import SyntheticHand21
    ds = SyntheticHand21(8, (128, 128), (32, 32))
    x, t, w, meta = ds[3]
    assert tuple(x.shape) == (3, 128, 128) and tuple(t.shape) == (21, 32, 32) and tuple(w.shape) == (21, 1)
    assert ds.num_keypoints == 21 and set(ds.keypoints_group) == {'MCP', 'PIP', 'DIP', 'fingertip', 'all'}
    assert float(t.max()) == 1.0 and torch.equal(ds[3][0], x)
    assert abs(ds.group_accuracy(list(range(21)))['all'] - 10.0) < 1e-9


@pytest.mark.gpu
def test_train_resume_and_test_cli_on_gpu(gpu, tmp_path):
    log = str(tmp_path / 'run')
    common = ['data/none', '-t', 'Hand3DStudio', '--synthetic', '-a', 'resnet18', '-b', '4', '-i', '6', '-p', '2', '-j', '0',
              '--pretrain_epochs', '1', '--log', log]
    env = dict(os.environ, PYTHONPATH=PKG)

    def run(script, extra):
        r = subprocess.run([sys.executable, os.path.join(PKG, script)] + common + extra, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        return r.stdout

    out = run('train1.py', ['--epochs', '1', '--pretrain', str(tmp_path / 'none.pth')])
    assert 'Start regression domain adaptation.' in out and 'Target(best)' in out
    ck_path = os.path.join(log, 'checkpoints', '0.pth')
    # best.pth is only written when the target accuracy beats 0 (train1.py:266-268), which 6 iterations may not reach
    assert os.path.exists(ck_path) and os.path.exists(os.path.join(log, 'checkpoints', 'model_ema.pth')) \
        and os.path.exists(os.path.join(log, 'checkpoints', 'pretrain.pth'))
    ck = torch.load(ck_path, map_location='cpu', weights_only=False)
    for k in ('model', 'optimizer_f', 'optimizer_h', 'optimizer_h_adv', 'lr_scheduler_f', 'lr_scheduler_h', 'lr_scheduler_h_adv', 'epoch', 'args'):
        assert k in ck, k                      # the keys test.py:192-201 of the reference requires
    assert len(ck['model']) == 222 and ck['epoch'] == 0
    assert len(ck['optimizer_f']['state']) > 0
    out = run('train1.py', ['--epochs', '2', '--resume', ck_path])
    assert 'Epoch: [1]' in out and 'Epoch: [0]' not in out
    out = run('test.py', ['--checkpoint', ck_path])
    assert 'Source:' in out and 'fingertip:' in out


def test_resnext_and_wide_resnet_constructors_keep_the_torchvision_state_dict_layout():
    """reference uda/model/resnet.py:124-183 (resnext50_32x4d, resnext101_32x8d, wide_resnet50_2, wide_resnet101_2 pass `groups` /
    `width_per_group` to torchvision's ResNet): parameter names and shapes as torchvision's (known shapes of its Bottleneck, v1.5)."""
    import uda.model as models
    assert {'resnext50_32x4d', 'resnext101_32x8d', 'wide_resnet50_2', 'wide_resnet101_2'} <= set(models.__all__)
    sd = models.resnext50_32x4d().state_dict()
    assert tuple(sd['layer1.0.conv1.weight'].shape) == (128, 64, 1, 1)
    assert tuple(sd['layer1.0.conv2.weight'].shape) == (128, 4, 3, 3)          # 32 groups x 4 channels
    assert tuple(sd['layer1.0.conv3.weight'].shape) == (256, 128, 1, 1)
    assert tuple(sd['layer4.2.conv2.weight'].shape) == (1024, 32, 3, 3)
    assert tuple(sd['layer4.0.downsample.0.weight'].shape) == (2048, 1024, 1, 1) and tuple(sd['fc.weight'].shape) == (1000, 2048)
    sd = models.resnext101_32x8d().state_dict()
    assert tuple(sd['layer1.0.conv2.weight'].shape) == (256, 8, 3, 3) and tuple(sd['layer3.22.conv2.weight'].shape) == (1024, 32, 3, 3)
    sd = models.wide_resnet50_2().state_dict()
    assert tuple(sd['layer1.0.conv2.weight'].shape) == (128, 128, 3, 3) and tuple(sd['layer4.2.conv1.weight'].shape) == (1024, 2048, 1, 1)
    assert tuple(sd['layer4.2.conv3.weight'].shape) == (2048, 1024, 1, 1)
    m = models.wide_resnet101_2()
    assert len(m.layer3) == 23 and m.out_features == 2048
    assert len(models.resnet50().state_dict()) == len(models.wide_resnet50_2().state_dict()) == len(models.resnext50_32x4d().state_dict())
    with pytest.raises(ValueError):
        models.ResNet(models.resnet.BasicBlock, [2, 2, 2, 2], groups=32, width_per_group=4)
