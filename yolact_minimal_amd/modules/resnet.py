"""ResNet-50/101 parameter container for the HIP engine.

The reference backbone is `/root/reference/modules/resnet.py:5-104`
(Bottleneck `:5-40`, ResNet `:43-104`).  Here the modules only *own parameters
and buffers* under the reference's state-dict key names
(`backbone.layers.L.B.{conv,bn}{1,2,3}`, `...0.downsample.{0,1}`,
`backbone.conv1/bn1`) and are created in the reference's order so that
`torch.manual_seed(s)` followed by construction gives bit-identical weights.
No arithmetic happens in these classes: `yolact_minimal_amd.engine` walks them,
folds BN into a per-channel scale/shift, repacks OIHW -> [Cout][kh][kw][Cin]
and launches the fused conv+BN+ReLU(+residual) HIP kernels.
"""
import torch
import torch.nn as nn


class Bottleneck(nn.Module):
    """1x1 -> 3x3(stride) -> 1x1(x4) residual block; stride sits on the 3x3 (resnet.py:12)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = norm_layer(planes * self.expansion)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError('Bottleneck is a parameter container; the HIP engine executes it '
                           '(yolact_minimal_amd.engine). There is no eager fallback.')


class ResNet(nn.Module):
    def __init__(self, layers, block=Bottleneck, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.num_base_layers = len(layers)
        # registered before conv1 on purpose: named_modules() order drives the xavier
        # re-initialisation order in Yolact.__init__ (reference resnet.py:49 vs :55).
        self.layers = nn.ModuleList()
        self.channels = []
        self.norm_layer = norm_layer
        self.inplanes = 64

        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)

        for planes, blocks, stride in zip((64, 128, 256, 512), layers, (1, 2, 2, 2)):
            self._make_layer(block, planes, blocks, stride)

        self.backbone_modules = [m for m in self.modules() if isinstance(m, nn.Conv2d)]

    def _make_layer(self, block, planes, blocks, stride):
        out_ch = planes * block.expansion
        downsample = None
        if stride != 1 or self.inplanes != out_ch:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, out_ch, 1, stride=stride, bias=False),
                self.norm_layer(out_ch))
        stage = [block(self.inplanes, planes, stride, downsample, self.norm_layer)]
        self.inplanes = out_ch
        stage += [block(out_ch, planes, norm_layer=self.norm_layer) for _ in range(1, blocks)]
        self.channels.append(out_ch)
        self.layers.append(nn.Sequential(*stage))

    def forward(self, x):
        raise RuntimeError('ResNet is a parameter container; use Yolact.forward (HIP engine).')

    def init_backbone(self, path):
        """Strict load of an ImageNet backbone checkpoint (reference resnet.py:100-104)."""
        state_dict = torch.load(path, map_location='cpu')
        self.load_state_dict(state_dict, strict=True)
        print(f'\nBackbone is initiated with {path}.\n')
