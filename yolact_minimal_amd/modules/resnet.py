"""ResNet-50/101 parameter container for the HIP engine.

The reference backbone is `/root/reference/modules/resnet.py:5-104` (Bottleneck `:5-40`, ResNet `:43-104`).  Nothing is
computed here: these modules only *own* the parameters and BatchNorm buffers under the reference's state-dict key names
(`backbone.conv1 / bn1`, `backbone.layers.<stage>.<block>.{conv,bn}{1,2,3}`, `...<block>.downsample.{0,1}`), registered in the
reference's order so that `torch.manual_seed(s)` + construction yields bit-identical weights
(`tests/test_oracle_golden.py::test_seeded_state_dict_matches_reference`).  `yolact_minimal_amd.engine` (inference) and
`yolact_minimal_amd.train_engine` (training) walk the containers and launch the fused conv + BN + ReLU (+ residual) kernels.

The network is described by a table — (bottleneck width, stride of the stage's first block) per stage — instead of being
spelled out call by call; a stage's first block gets a projection shortcut whenever its input and output shapes differ.
"""
import torch
import torch.nn as nn

STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))      # (bottleneck width, stride of the first block), resnet.py:59-62
STEM_WIDTH = 64
_NOT_EXECUTABLE = ('{} only holds parameters: the HIP engine executes the network (Yolact.forward); '
                   'yolact_minimal_amd has no eager fallback.')


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)


class Bottleneck(nn.Module):
    """1x1 reduce -> 3x3 (carries the stride, resnet.py:12) -> 1x1 expand (x4), plus the shortcut."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=nn.BatchNorm2d):
        super().__init__()
        wide = planes * self.expansion
        # (name suffix, conv, channels of its norm) in registration order: conv1 bn1 conv2 bn2 conv3 bn3
        for n, conv, ch in ((1, _conv(inplanes, planes, 1), planes), (2, _conv(planes, planes, 3, stride), planes),
                            (3, _conv(planes, wide, 1), wide)):
            self.add_module(f'conv{n}', conv)
            self.add_module(f'bn{n}', norm_layer(ch))
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError(_NOT_EXECUTABLE.format('Bottleneck'))


class ResNet(nn.Module):
    def __init__(self, layers, block=Bottleneck, norm_layer=nn.BatchNorm2d):
        super().__init__()
        # `layers` is registered before the stem on purpose: named_modules() order is the order in which Yolact.__init__
        # re-draws the conv weights (reference resnet.py:49 vs :55), i.e. it is part of the seeded-init contract.
        self.layers = nn.ModuleList()
        self.conv1 = nn.Conv2d(3, STEM_WIDTH, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(STEM_WIDTH)
        self.channels = []
        width_in = STEM_WIDTH
        for (planes, stride), depth in zip(STAGES, layers):
            width_out = planes * block.expansion
            shortcut = None
            if stride != 1 or width_in != width_out:            # projection shortcut, created before the block's own convs
                shortcut = nn.Sequential(_conv(width_in, width_out, 1, stride), norm_layer(width_out))
            blocks = [block(width_in, planes, stride, shortcut, norm_layer)]
            blocks.extend(block(width_out, planes, norm_layer=norm_layer) for _ in range(depth - 1))
            self.layers.append(nn.Sequential(*blocks))
            self.channels.append(width_out)
            width_in = width_out

    def forward(self, x):
        raise RuntimeError(_NOT_EXECUTABLE.format('ResNet'))

    def init_backbone(self, path):
        """Strict load of an ImageNet backbone checkpoint (reference resnet.py:100-104)."""
        self.load_state_dict(torch.load(path, map_location='cpu'), strict=True)
        print(f'\nBackbone is initiated with {path}.\n')
