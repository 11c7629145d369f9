"""U-Net / dual-decoder U-Net with the reference's module tree and state-dict contract, executed on libmseg_hip.

Mirror of ``src/utils/unets.py`` (reference): ``build_unet`` (:8-57), ``get_weights`` (:60-78), ``Mish`` (:81-89),
``ConvBlock`` (:92-173), ``ConvPool`` (:176-226), ``TranspConvBlock`` (:229-264), ``UNet`` (:267-377),
``DUNet`` (:380-506).  The ``nn.Module`` tree only *holds parameters* (same names, shapes, default initialisation
and construction order as the reference, so checkpoints and seeded inits are interchangeable); ``forward`` hands the
whole network to ``microbeseg_amd.engine`` which runs hand-written HIP kernels.  Sub-blocks are not individually
callable: there is no eager PyTorch compute path.
"""
import torch
import torch.nn as nn

from .. import engine

_ACTS = ("relu", "leakyrelu", "elu", "mish")
_NORMS = ("bn", "gn", "in")


class Mish(nn.Module):
    """x * tanh(softplus(x)) — marker module; the maths lives in csrc/common.h (act_fwd / act_bwd)."""


def _activation(act_fun):
    if act_fun == 'relu':
        return nn.ReLU(inplace=True)
    if act_fun == 'leakyrelu':
        return nn.LeakyReLU(inplace=True)
    if act_fun == 'elu':
        return nn.ELU(inplace=True)
    if act_fun == 'mish':
        return Mish()
    raise Exception('Unsupported activation function: {}'.format(act_fun))


def _normalization(normalization, channels):
    if normalization == 'bn':
        return nn.BatchNorm2d(channels)
    if normalization == 'gn':
        return nn.GroupNorm(num_groups=8, num_channels=channels)
    if normalization == 'in':
        return nn.InstanceNorm2d(num_features=channels)
    raise Exception('Unsupported normalization: {}'.format(normalization))


class _ParamBlock(nn.Module):
    def forward(self, *args, **kwargs):
        raise RuntimeError("microbeseg_amd blocks are parameter holders; call the enclosing UNet/DUNet")


class ConvBlock(_ParamBlock):
    """[0] conv3x3 [1] act [2] norm [3] conv3x3 [4] act [5] norm — state-dict slots as in the reference."""

    def __init__(self, ch_in, ch_out, act_fun, normalization):
        super().__init__()
        layers = []
        for cin in (ch_in, ch_out):
            layers.append(nn.Conv2d(cin, ch_out, kernel_size=3, stride=1, padding=1, bias=True))
            layers.append(_activation(act_fun))
            layers.append(_normalization(normalization, ch_out))
        self.conv = nn.Sequential(*layers)

    def specs(self, act_fun, normalization):
        return (engine.ConvSpec("conv", self.conv[0], self.conv[2], act_fun, normalization),
                engine.ConvSpec("conv", self.conv[3], self.conv[5], act_fun, normalization))


class ConvPool(_ParamBlock):
    """[0] conv3x3 stride 2 [1] act [2] norm."""

    def __init__(self, ch_in, act_fun, normalization):
        super().__init__()
        self.conv_pool = nn.Sequential(nn.Conv2d(ch_in, ch_in, kernel_size=3, stride=2, padding=1, bias=True),
                                       _activation(act_fun), _normalization(normalization, ch_in))

    def spec(self, act_fun, normalization):
        return engine.ConvSpec("pool", self.conv_pool[0], self.conv_pool[2], act_fun, normalization)


class TranspConvBlock(_ParamBlock):
    """up = ConvTranspose2d(2, stride 2) followed by norm (no activation)."""

    def __init__(self, ch_in, ch_out, normalization):
        super().__init__()
        self.up = nn.Sequential(nn.ConvTranspose2d(ch_in, ch_out, kernel_size=2, stride=2))
        self.norm = _normalization(normalization, ch_out)

    def spec(self, normalization):
        return engine.ConvSpec("up", self.up[0], self.norm, "none", normalization)


class _HipNet(nn.Module):
    n_decoders = 1

    def __init__(self, ch_in, ch_outs, pool_method, act_fun, normalization, filters):
        super().__init__()
        if act_fun not in _ACTS:
            raise Exception('Unsupported activation function: {}'.format(act_fun))
        if normalization not in _NORMS:
            raise Exception('Unsupported normalization: {}'.format(normalization))
        self.ch_in = ch_in
        self.filters = filters
        self.pool_method = pool_method
        self.act_fun = act_fun
        self.normalization = normalization

        # encoder — construction order matches the reference so that a seeded default init is identical
        self.encoderConv = nn.ModuleList()
        if pool_method == 'max':
            self.pooling = nn.MaxPool2d(kernel_size=2, stride=2)
        elif pool_method == 'conv':
            self.pooling = nn.ModuleList()
        n = filters[0]
        self.encoderConv.append(ConvBlock(ch_in, n, act_fun, normalization))
        if pool_method == 'conv':
            self.pooling.append(ConvPool(n, act_fun, normalization))
        while n < filters[1]:
            self.encoderConv.append(ConvBlock(n, n * 2, act_fun, normalization))
            if n * 2 < filters[1] and pool_method == 'conv':
                self.pooling.append(ConvPool(n * 2, act_fun, normalization))
            n *= 2

        ups, convs = self._decoder_lists()
        while n > filters[0]:
            for up, conv in zip(ups, convs):
                up.append(TranspConvBlock(n, n // 2, normalization))
                conv.append(ConvBlock(n, n // 2, act_fun, normalization))
            n //= 2
        for conv, co in zip(convs, ch_outs):
            conv.append(nn.Conv2d(n, co, kernel_size=1, stride=1, padding=0))

        self._spec_cache = None
        self._ws = {}

    # -- engine glue -------------------------------------------------------------------------------------------
    @property
    def _spec(self):
        if self._spec_cache is None:
            a, nm = self.act_fun, self.normalization
            enc = []
            nlev = len(self.encoderConv)
            for i, blk in enumerate(self.encoderConv):
                c1, c2 = blk.specs(a, nm)
                pool = None
                if i < nlev - 1:
                    pool = self.pooling[i].spec(a, nm) if self.pool_method == 'conv' else 'max'
                enc.append(dict(c1=c1, c2=c2, pool=pool))
            decs = []
            for ups, convs in zip(*self._decoder_lists()):
                levels = []
                for up, blk in zip(ups, list(convs)[:-1]):
                    c1, c2 = blk.specs(a, nm)
                    levels.append(dict(up=up.spec(nm), c1=c1, c2=c2))
                decs.append(dict(levels=levels, head=engine.HeadSpec(convs[-1])))
            object.__setattr__(self, "_spec_cache", engine.NetSpec(self.ch_in, enc, decs, self.pool_method))
        return self._spec_cache

    def _workspace(self, device):
        key = str(device)
        if key not in self._ws:
            self._ws[key] = engine.Workspace(device)
        return self._ws[key]

    def _run(self, x):
        return engine.run_module(self, x)


class UNet(_HipNet):
    """U-Net (Ronneberger et al. 2015 with zero padding, norm layers, transposed-conv upsampling)."""

    def __init__(self, ch_in=1, ch_out=1, pool_method='conv', act_fun='relu', normalization='bn', filters=(64, 1024)):
        self.decoderUpconv = None
        super().__init__(ch_in, (ch_out,), pool_method, act_fun, normalization, filters)
        self.ch_out = ch_out

    def _decoder_lists(self):
        if self.decoderUpconv is None:
            self.decoderUpconv = nn.ModuleList()
            self.decoderConv = nn.ModuleList()
        return (self.decoderUpconv,), (self.decoderConv,)

    def forward(self, x):
        return self._run(x)[0]


class DUNet(_HipNet):
    """U-Net with a shared encoder and two decoder paths (decoder 1: borders/seeds, decoder 2: cells)."""

    def __init__(self, ch_in=1, ch_out=1, pool_method='conv', act_fun='relu', normalization='bn', filters=(64, 1024)):
        self.decoder1Upconv = None
        super().__init__(ch_in, (ch_out, 1), pool_method, act_fun, normalization, filters)

    def _decoder_lists(self):
        if self.decoder1Upconv is None:
            self.decoder1Upconv = nn.ModuleList()
            self.decoder1Conv = nn.ModuleList()
            self.decoder2Upconv = nn.ModuleList()
            self.decoder2Conv = nn.ModuleList()
        return (self.decoder1Upconv, self.decoder2Upconv), (self.decoder1Conv, self.decoder2Conv)

    def forward(self, x):
        x1, x2 = self._run(x)
        return x1, x2


def build_unet(unet_type, act_fun, pool_method, normalization, device, num_gpus, ch_in=1, ch_out=1, filters=(64, 1024)):
    """ Build U-net architecture (same signature as the reference, src/utils/unets.py:8).

    :param unet_type: 'U' (U-net) or 'DU' (U-net with two decoder paths and two outputs).
    :param act_fun: 'relu', 'leakyrelu', 'elu', 'mish' (not in the output layer).
    :param pool_method: 'max' (maximum pooling), 'conv' (convolution with stride 2).
    :param normalization: 'bn', 'gn' (8 groups), 'in'.
    :param device: torch device ('cuda[:N]'; ROCm PyTorch keeps the cuda device name).
    :param num_gpus: > 1 wraps the model for data-parallel training (one process per GPU, RCCL gradient all-reduce;
        replaces the reference's single-process nn.DataParallel, unets.py:51-52).  ``.module`` holds the bare model.
    :return: model
    """
    if unet_type == 'DU':
        model = DUNet(ch_in=ch_in, ch_out=ch_out, pool_method=pool_method, filters=filters, act_fun=act_fun,
                      normalization=normalization)
    elif unet_type == 'U':
        model = UNet(ch_in=ch_in, ch_out=ch_out, pool_method=pool_method, filters=filters, act_fun=act_fun,
                     normalization=normalization)
    else:
        raise Exception('Architecture "{}" is not known'.format(unet_type))

    if num_gpus > 1:
        from ..parallel import RcclDataParallel
        model = RcclDataParallel(model)

    model = model.to(device)
    return model


def get_weights(net, weights, device, num_gpus):
    """ Load a reference-format ``.pth`` state dict into the model (src/utils/unets.py:60-78). """
    state = torch.load(weights, map_location=device)
    if num_gpus > 1:
        net.module.load_state_dict(state)
    else:
        net.load_state_dict(state)
    return net
