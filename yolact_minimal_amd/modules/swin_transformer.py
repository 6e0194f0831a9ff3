"""Swin-T parameter container for the HIP engine.

Reference: `/root/reference/modules/swin_transformer.py` — PatchEmbed `:400-433`, BasicLayer `:328-397`,
SwinTransformerBlock `:203-289`, WindowAttention `:131-200`, Mlp `:83-96`, PatchMerging `:292-325`, SwinTransformer
`:436-518`.  As with the ResNet container, these classes only own parameters under the reference's state-dict names
(`patch_embed.{proj,norm}`, `layers.L.blocks.B.{norm1,attn.{relative_position_bias_table,relative_position_index,qkv,
proj},norm2,mlp.{fc1,fc2}}`, `layers.L.downsample.{reduction,norm}`, `norm{1,2,3}`) and are constructed in the
reference's order with the same initialisers, so seeded construction is bit-identical.  The arithmetic is in
`yolact_minimal_amd/engine.py` (LayerNorm / window-attention / patch-merge HIP kernels + the MFMA GEMM).
"""
import math

import torch
import torch.nn as nn


def _trunc_normal_(t, std=1., a=-2., b=2.):
    # same sequence of in-place ops (and RNG draws) as the reference helper (:9-59), mean = 0
    def cdf(x):
        return (1. + math.erf(x / math.sqrt(2.))) / 2.
    with torch.no_grad():
        lo, hi = cdf(a / std), cdf(b / std)
        t.uniform_(2 * lo - 1, 2 * hi - 1)
        t.erfinv_()
        t.mul_(std * math.sqrt(2.))
        t.add_(0.)
        t.clamp_(min=a, max=b)
    return t


def relative_position_index(ws):
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing='ij')).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        self.register_buffer('relative_position_index', relative_position_index(window_size))
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        _trunc_normal_(self.relative_position_bias_table, std=.02)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4., drop_path=0.):
        super().__init__()
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, window_size, shift_size
        self.drop_prob = float(drop_path)           # DropPath rate of this block (train mode only, reference :226,287-288)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4., downsample=False, drop_path=None):
        super().__init__()
        self.window_size, self.shift_size, self.depth = window_size, window_size // 2, depth
        drop_path = drop_path or [0.] * depth
        self.blocks = nn.ModuleList(SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2,
                                                         mlp_ratio, drop_path[i]) for i in range(depth))
        self.downsample = PatchMerging(dim) if downsample else None


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96):
        super().__init__()
        self.patch_size, self.in_chans, self.embed_dim = (patch_size, patch_size), in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dim)


class SwinTransformer(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=7,
                 mlp_ratio=4., drop_path_rate=0.2):
        super().__init__()
        self.num_layers, self.embed_dim, self.depths, self.heads = len(depths), embed_dim, depths, num_heads
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]      # stochastic depth decay rule (:464)
        self.window_size = window_size
        self.out_norm_indices = (1, 2, 3)
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim)
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i], window_size, mlp_ratio,
                                          downsample=i < self.num_layers - 1,
                                          drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])]))
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in self.out_norm_indices:
            self.add_module(f'norm{i}', nn.LayerNorm(self.num_features[i]))

    def forward(self, x):
        raise RuntimeError('SwinTransformer is a parameter container; use Yolact.forward (HIP engine).')

    def init_backbone(self, weight):
        """Reference `:486-498`: re-initialise Linear / LayerNorm, then non-strict load of the ImageNet checkpoint."""
        def _init(m):
            if isinstance(m, nn.Linear):
                _trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        self.apply(_init)
        self.load_state_dict(torch.load(weight, map_location='cpu'), strict=False)
        print(f'\nBackbone is initiated with {weight}.\n')
