"""Every kernel-selection switch that still ships is a code path the default tests do not run: this module runs the golden
model-level parity tests (G1 neck + heads, G7 two full A/B/C iterations and -- where noted -- G8, the ResNet-50 iteration, against the
reference-captured fixtures; the skip-discarded-work mode the training step uses) once under each of them.  The switches are read when the library / package is first imported, hence one subprocess per variant."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

G8 = ' or (g8_resnet50_iteration and True)'
VARIANTS = [
    ({'MI355_PGEMM': '0'}, G8),              # 1x1 convs on the gather kernel only
    ({'MI355_PGEMM': '2'}, G8),              # weights-stationary streaming GEMM wherever it fits (also the small maps, residual / accumulate epilogues)
    ({'MI355_CAT': '0'}, G8),                # fusion heads: heat-map conv + feature conv as two launches
    ({'MI355_STEM_S2D': '0'}, G8),           # stem as the 7x7 / stride-2 conv over the channel-padded image (default: 4x4 over the folded one)
    ({'MI355_BN_BWD_FUSE': '1'}, G8),        # BatchNorm-backward reduction in the dgrad epilogue (EPI = 2 of the gather kernel)
    # stand-alone statistics pass (moves G8's step-B loss 5.4e-3 from fp64, as any other re-rounding of the forward does on that
    # ill-conditioned random-init ResNet-50: profiles/r04_stem_s2d.txt section 4)
    ({'MI355_BN_STATS_FUSE': '0'}, G8),
    ({'MI355_BN_RESIDENT': '0'}, G8),        # three-launch BatchNorm backward
    ({'MI355_WGRAD_GROUP': '0'}, G8),        # every weight gradient launched on its own
    ({'MI355_WGRAD_KW': '0', 'MI355_WGRAD_KW2': '0', 'MI355_KW3': '0', 'MI355_DMA': '0'}, G8),      # generic kernels everywhere
    ({'MI355_BN_LAZY_DRES': '0', 'MI355_SKIP_FUSE': '0'}, G8),                                        # host-side fusions off
    ({'MI355_ZERO_BN_BIAS_GRAD': '0'}, G8),   # bias gradients in front of a BatchNorm as column sums (rounding noise instead of an exact zero)
]


@pytest.mark.parametrize('env,extra', VARIANTS, ids=lambda e: ','.join('%s=%s' % kv for kv in e.items()) if isinstance(e, dict) else ('+g8' if e else ''))
def test_golden_model_parity_under_switch(gpu, env, extra):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_model.py'), '-x', '-q', '-m', 'gpu',
                        '-k', 'g1_neck or (g7_full_iteration and True)' + extra], capture_output=True, text=True, env=e, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout, r.stdout[-2000:]
