"""CPU oracle for the MoPoE joint-ELBO hot path.  TEST INFRASTRUCTURE ONLY.

This file is a *restatement* (plain PyTorch on CPU, functional style, fp32 or fp64) of the
algorithm the reference runs in ``mimic/run_epochs.py:52-96`` (``basic_routine_epoch``) and below.
It is the checker the HIP product path is compared against; nothing under ``mopoe-mimic_amd/`` may
import it.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it.

Parity status: PINNED.  The reference has no numeric known-answer tests for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself, generated in the
build container by ``oracle/gen_golden.py`` (imports /root/reference) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every vector.

All tensors use the reference's layouts (NCHW / NCL) and the reference's ``state_dict`` key names
(SURVEY.md Appendix B), so a reference checkpoint can be fed in directly.

Reference citations (relative to /root/reference):
  image encoder        mimic/networks/ConvNetworksImgMimic.py:20-36, FeatureExtractorImg.py:23-81
  image decoder        mimic/networks/ConvNetworksImgMimic.py:39-54, DataGeneratorImg.py:29-98
  residual blocks      mimic/networks/ResidualBlocks.py:5-131
  text encoder         mimic/networks/ConvNetworksTextMimic.py:11-36, word_encoding/mmvae_text_enc.py:22-85
  text decoder         mimic/networks/ConvNetworksTextMimic.py:39-68, word_encoding/DataGeneratorText.py:29-98
  char text networks   mimic/networks/char_encoding/FeatureExtractorText.py:28-81, char_encoding/DataGeneratorText.py:25-76
                       (text_encoding='char': [B, 1024, 71] inputs, 8 residual blocks each way, ConvTranspose1d head)
  latent compressor    mimic/networks/FeatureCompressor.py:4-28
  subset PoE + MoE     mimic/utils/BaseMMVae.py:139-196, evaluation/divergence_measures/mm_div.py:10-17
  mixture selection    mimic/utils/utils.py:51-77
  KL                   mimic/evaluation/divergence_measures/kl_div.py:8-16, mm_div.py:90-110
  reparameterise       mimic/utils/utils.py:45-48
  likelihoods          mimic/modalities/Modality.py:25-30, MimicText.py:37-40, modalities/utils.py:4-15
  loss assembly        mimic/evaluation/losses.py:6-31,80-89
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from itertools import combinations
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

MOD_ORDER = ("PA", "Lateral", "text")  # dict order of exp.modalities (experiment.py:80-92)
RES_A, RES_B = 2.0, 0.3  # residual mix (FeatureExtractorImg.py:24, DataGeneratorImg.py:30)
BN_EPS = 1e-5
LAPLACE_SCALE = 0.75  # ConvNetworksImgMimic.py:54
POE_EPS = 1e-8  # mm_div.py:10


@dataclass
class Cfg:
    img_size: int = 128
    class_dim: int = 128
    DIM_img: int = 64
    DIM_text: int = 128
    vocab_size: int = 3517
    len_sequence: int = 128
    image_channels: int = 1
    text_encoding: str = "word"  # 'word' (ids [B, L]) or 'char' (dense / one-hot [B, 1024, num_features], flags.py:48,157)
    num_features: int = 71       # alphabet size of the char encoding (experiment.py:49-51)
    batch_size: int = 64  # flags.batch_size: the NORMALISER (kl_div.py:14-15, Modality.py:30)
    beta: float = 1.0
    beta_content: float = 1.0
    rec_weights: Dict[str, float] = field(default_factory=lambda: {"PA": 0.33, "Lateral": 0.33, "text": 0.33})


# --------------------------------------------------------------------------------------------
# architecture tables (channel plans), shared by init_state and the forward functions
# --------------------------------------------------------------------------------------------
def img_enc_blocks(cfg: Cfg) -> List[Tuple[int, int, int, int]]:
    """(cin, cout, stride, pad) of resblock_1.. (FeatureExtractorImg.py:35-59); kernel is always 4."""
    d = cfg.DIM_img
    blocks = [(d, 2 * d, 2, 1), (2 * d, 3 * d, 2, 1), (3 * d, 4 * d, 2, 1)]
    if cfg.img_size == 64:
        blocks += [(4 * d, 5 * d, 2, 0)]
    elif cfg.img_size == 128:
        blocks += [(4 * d, 5 * d, 2, 1), (5 * d, 5 * d, 2, 0)]
    elif cfg.img_size == 256:
        blocks += [(4 * d, 5 * d, 4, 1), (5 * d, 5 * d, 2, 0)]
    else:
        raise ValueError("img_size must be 64, 128 or 256 (FeatureExtractorImg.py:41-59)")
    return blocks


def img_dec_blocks(cfg: Cfg) -> List[Tuple[int, int, int, int]]:
    """(cin, cout, stride, pad) of generator.0.. (DataGeneratorImg.py:33-82); kernel 4."""
    d = cfg.DIM_img
    blocks = [(5 * d, 4 * d, 1, 0), (4 * d, 3 * d, 2, 1), (3 * d, 2 * d, 2, 1), (2 * d, d, 2, 1)]
    if cfg.img_size == 128:
        blocks += [(d, d, 2, 1)]
    if cfg.img_size == 256:
        blocks += [(d, d, 2, 1), (d, d, 2, 1)]
    return blocks


def text_enc_blocks(cfg: Cfg) -> List[Tuple[int, int, int, int]]:
    """all 8 allocated blocks (mmvae_text_enc.py:32-56); only the first 6 run for len_sequence<=500."""
    d = cfg.DIM_text
    return [(d, 2 * d, 2, 1), (2 * d, 3 * d, 2, 1), (3 * d, 4 * d, 2, 1), (4 * d, 4 * d, 2, 1),
            (4 * d, 4 * d, 2, 1), (4 * d, 5 * d, 2, 1), (5 * d, 5 * d, 2, 1), (5 * d, 5 * d, 2, 0)]


def text_dec_blocks(cfg: Cfg) -> List[Tuple[int, int, int, int]]:
    """word/len-128 plan (word_encoding/DataGeneratorText.py:33-68); char/len-1024 plan
    (char_encoding/DataGeneratorText.py:28-50: resblock_1..8, then the ConvTranspose1d head conv2)."""
    d = cfg.DIM_text
    if cfg.text_encoding == "char":
        if cfg.len_sequence != 1024:
            raise NotImplementedError("the char networks only close for len_sequence = 1024 (flags.py:157)")
        return [(5 * d, 5 * d, 1, 0), (5 * d, 5 * d, 2, 1), (5 * d, 5 * d, 2, 1), (5 * d, 4 * d, 2, 1),
                (4 * d, 4 * d, 2, 1), (4 * d, 3 * d, 2, 1), (3 * d, 2 * d, 2, 1), (2 * d, d, 2, 1)]
    if cfg.len_sequence != 128:
        raise NotImplementedError("only the word / len_sequence=128 path is in scope (SURVEY §2.1-7)")
    return [(5 * d, 5 * d, 1, 0), (5 * d, 5 * d, 2, 1), (5 * d, 5 * d, 2, 1), (5 * d, 4 * d, 2, 1),
            (4 * d, 4 * d, 2, 1), (4 * d, d, 2, 1)]


def subset_table() -> List[Tuple[str, Tuple[str, ...]]]:
    """non-empty subsets in the reference's dict order with members sorted by name
    (BaseExperiment.py:66-82)."""
    out = []
    for n in range(1, len(MOD_ORDER) + 1):
        for combo in combinations(MOD_ORDER, n):
            members = tuple(sorted(combo))
            out.append(("_".join(members), members))
    return out


# --------------------------------------------------------------------------------------------
# seeded weights (the BUILD's generator; loaded into the reference by gen_golden.py)
# --------------------------------------------------------------------------------------------
def init_state(cfg: Cfg, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Deterministic state_dict with the reference key names / shapes.
    Conv/linear weights ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)); BN affine perturbed away from (1,0) so
    that tests exercise gamma/beta; running stats perturbed so eval mode is exercised."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def uni(shape, bound):
        return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1).mul_(bound).to(dtype)

    def conv(key, cout, cin, *k, bias=True, transposed=False):
        fan_in = (cout if transposed else cin) * math.prod(k)
        shape = (cin, cout, *k) if transposed else (cout, cin, *k)
        sd[key + ".weight"] = uni(shape, 1.0 / math.sqrt(fan_in))
        if bias:
            sd[key + ".bias"] = uni((cout,), 1.0 / math.sqrt(fan_in))

    def bn(key, c):
        sd[key + ".weight"] = (1.0 + uni((c,), 0.3)).to(dtype)
        sd[key + ".bias"] = uni((c,), 0.2)
        sd[key + ".running_mean"] = uni((c,), 0.1)
        sd[key + ".running_var"] = (1.0 + uni((c,), 0.2)).to(dtype)
        sd[key + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    def linear(key, cout, cin):
        sd[key + ".weight"] = uni((cout, cin), 1.0 / math.sqrt(cin))
        sd[key + ".bias"] = uni((cout,), 1.0 / math.sqrt(cin))

    def resblock(prefix, cin, cout, k, transposed, bias_main, short_name):
        conv(prefix + ".conv1", cin, cin, *([1] * len(k)), bias=bias_main, transposed=transposed)
        bn(prefix + ".bn1", cin)
        bn(prefix + ".bn2", cin)
        conv(prefix + ".conv2", cout, cin, *k, bias=bias_main, transposed=transposed)
        conv(prefix + f".{short_name}.0", cout, cin, *k, bias=True, transposed=transposed)
        bn(prefix + f".{short_name}.1", cout)

    for enc, dec in (("encoder_pa", "decoder_pa"), ("encoder_lat", "decoder_lat")):
        conv(f"{enc}.feature_extractor.conv1", cfg.DIM_img, cfg.image_channels, 3, 3, bias=False)
        for i, (ci, co, _s, _p) in enumerate(img_enc_blocks(cfg)):
            resblock(f"{enc}.feature_extractor.resblock_{i + 1}.0", ci, co, (4, 4), False, False, "downsample")
        linear(f"{enc}.feature_compressor.content_mu", cfg.class_dim, 5 * cfg.DIM_img)
        linear(f"{enc}.feature_compressor.content_logvar", cfg.class_dim, 5 * cfg.DIM_img)
        linear(f"{dec}.feature_generator", 5 * cfg.DIM_img, cfg.class_dim)
        blocks = img_dec_blocks(cfg)
        for i, (ci, co, _s, _p) in enumerate(blocks):
            resblock(f"{dec}.img_generator.generator.{i}.0", ci, co, (4, 4), True, False, "upsample")
        conv(f"{dec}.img_generator.generator.{len(blocks)}", cfg.image_channels, cfg.DIM_img, 3, 3,
             bias=True, transposed=True)

    d = cfg.DIM_text
    char = cfg.text_encoding == "char"
    if char:   # no embedding: the stem convolves the [B, L, num_features] input itself
        conv("encoder_text.feature_extractor.conv1", d, cfg.num_features, 4, bias=True)
    else:
        sd["encoder_text.feature_extractor.embedding.weight"] = torch.randn(
            (cfg.vocab_size, d), generator=g, dtype=torch.float64).to(dtype)
        sd["encoder_text.feature_extractor.embedding.weight"][0].zero_()  # padding_idx=0 (mmvae_text_enc.py:27)
        conv("encoder_text.feature_extractor.conv1", d, d, 4, bias=True)
    for i, (ci, co, _s, _p) in enumerate(text_enc_blocks(cfg)):
        resblock(f"encoder_text.feature_extractor.resblock_{i + 1}.0", ci, co, (4,), False, True, "downsample")
    linear("encoder_text.feature_compressor.content_mu", cfg.class_dim, 5 * d)
    linear("encoder_text.feature_compressor.content_logvar", cfg.class_dim, 5 * d)
    linear("decoder_text.feature_generator", 5 * d, cfg.class_dim)
    tblocks = text_dec_blocks(cfg)
    if char:   # char_encoding/DataGeneratorText.py: resblock_1..8 + conv2 (ConvTranspose1d d -> num_features, k4 s2 p1)
        for i, (ci, co, _s, _p) in enumerate(tblocks):
            resblock(f"decoder_text.text_generator.resblock_{i + 1}.0", ci, co, (4,), True, True, "upsample")
        conv("decoder_text.text_generator.conv2", cfg.num_features, d, 4, bias=True, transposed=True)
    else:
        for i, (ci, co, _s, _p) in enumerate(tblocks):
            resblock(f"decoder_text.text_generator.generator.{i}.0", ci, co, (4,), True, True, "upsample")
        conv(f"decoder_text.text_generator.generator.{len(tblocks)}", cfg.vocab_size, d, 1, bias=True)
    return sd


# --------------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------------
class Ctx:
    """Execution context: mode + injected dropout masks + (optional) capture of new BN running stats."""

    def __init__(self, mode: str = "train_nodrop", masks: Optional[Dict[str, torch.Tensor]] = None,
                 draw_masks: bool = False, record_masks: bool = False, mask_seed: Optional[int] = None,
                 bf16: bool = False):
        assert mode in ("train", "train_nodrop", "eval")
        self.mode = mode
        self.masks = masks if masks is not None else {}
        self.draw_masks = draw_masks  # the CPU timing baseline, and tests that need masks at sizes no fixture holds
        self.record_masks = record_masks  # keep the drawn masks in self.masks (replayed into the HIP path by tests)
        self.mask_gen = torch.Generator().manual_seed(mask_seed) if mask_seed is not None else None
        # bf16 = the storage/operand rounding of the HIP bf16 path (BASELINE configs #3, #5): every activation the
        # kernels keep in HBM and every MFMA operand (activations after BN+ReLU, conv/linear weights) is rounded to
        # bfloat16, all sums stay fp32 -- the reference arithmetic with the product's rounding points, see _q()
        self.bf16 = bf16
        self.new_running: Dict[str, torch.Tensor] = {}


class _RoundBf16(torch.autograd.Function):
    """x -> bf16(x) (round to nearest even) kept in the working dtype; the gradient passing back through the same
    point is rounded too (the HIP bf16 path stores activation gradients in bf16 at the same places)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _GradRoundBf16(torch.autograd.Function):
    """identity whose gradient is rounded to bf16 (a gradient tensor the bf16 path stores, of an fp32 activation)"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _q(x, ctx: "Ctx"):
    return _RoundBf16.apply(x) if ctx.bf16 else x


def _qw(w, ctx: "Ctx"):
    """weights as the bf16 MFMA operand (the fp32 master copy receives the unrounded gradient)"""
    if not ctx.bf16:
        return w
    wd = w.detach()
    return w + (wd.to(torch.bfloat16).to(w.dtype) - wd)   # value: bf16(w); gradient: straight through to w


def _bn(sd, key, x, ctx: Ctx):
    """BatchNorm{1,2}d: batch stats + biased var in train modes, running stats in eval
    (torch.nn semantics the reference relies on; SURVEY §8c 'Third-party arithmetic')."""
    w, b = sd[key + ".weight"], sd[key + ".bias"]
    dims = [0] + list(range(2, x.dim()))
    shape = [1, -1] + [1] * (x.dim() - 2)
    if ctx.mode == "eval":
        mean, var = sd[key + ".running_mean"], sd[key + ".running_var"]
    else:
        mean = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False)
        n = x.numel() // x.shape[1]
        with torch.no_grad():
            ctx.new_running[key + ".running_mean"] = 0.9 * sd[key + ".running_mean"] + 0.1 * mean
            ctx.new_running[key + ".running_var"] = (0.9 * sd[key + ".running_var"]
                                                     + 0.1 * var * (n / max(n - 1, 1)))
    xhat = (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + BN_EPS)
    return xhat * w.view(shape) + b.view(shape)


def _drop(x, key, ctx: Ctx, channelwise: bool):
    """Dropout2d(p=.5) (2-D blocks) / Dropout(p=.5) (1-D blocks) as a multiplier tensor with values
    {0, 2} (ResidualBlocks.py:10,15,73,79,105,111)."""
    if ctx.mode != "train":
        return x
    if key in ctx.masks:
        return x * ctx.masks[key].to(x.dtype)
    if ctx.draw_masks:
        shape = (x.shape[0], x.shape[1], 1, 1) if channelwise else x.shape
        m = (torch.rand(shape, generator=ctx.mask_gen) < 0.5).to(x.dtype) * 2.0
        if ctx.record_masks:
            ctx.masks[key] = m
        return x * m
    raise KeyError(f"train mode needs a dropout mask for {key}")


def _resblock(sd, prefix, x, ctx: Ctx, *, stride, pad, transposed, twod, short_name):
    """One residual block: out = 2.0*BN(shortcut_conv(x)) + 0.3*main(x) (ResidualBlocks.py:20-33,
    51-65,84-97,118-131)."""
    if twod:
        convf = F.conv_transpose2d if transposed else F.conv2d
    else:
        convf = F.conv_transpose1d if transposed else F.conv1d
    # ctx.bf16: x arrives rounded (it is a stored tensor); the conv operands relu(bn(.)) and the weights are rounded
    # for the MFMA; d1 = drop1(conv1), s = conv_s and the block output are stored tensors (drop2(conv2) is consumed by
    # the residual mix in conv2's epilogue, in fp32)
    h = _q(F.relu(_bn(sd, prefix + ".bn1", x, ctx)), ctx)
    h = convf(h, _qw(sd[prefix + ".conv1.weight"], ctx), sd.get(prefix + ".conv1.bias"))
    h = _q(_drop(h, prefix + ".dropout1", ctx, twod), ctx)
    h = _q(F.relu(_bn(sd, prefix + ".bn2", h, ctx)), ctx)
    h = convf(h, _qw(sd[prefix + ".conv2.weight"], ctx), sd.get(prefix + ".conv2.bias"), stride=stride, padding=pad)
    h = _drop(h, prefix + ".dropout2", ctx, twod)
    s = _q(convf(x, _qw(sd[prefix + f".{short_name}.0.weight"], ctx), sd[prefix + f".{short_name}.0.bias"],
                 stride=stride, padding=pad), ctx)
    s = _bn(sd, prefix + f".{short_name}.1", s, ctx)
    return _q(RES_A * s + RES_B * h, ctx)


def _compress(sd, prefix, feats, ctx: Optional[Ctx] = None):
    ctx = ctx or Ctx()
    feats = feats.reshape(feats.shape[0], -1)
    # ctx.bf16: bf16 operands, fp32 results (mu / logvar feed the fp32 latent kernel unrounded)
    mu = F.linear(feats, _qw(sd[prefix + ".content_mu.weight"], ctx), sd[prefix + ".content_mu.bias"])
    lv = F.linear(feats, _qw(sd[prefix + ".content_logvar.weight"], ctx), sd[prefix + ".content_logvar.bias"])
    return mu, lv


def encode_img(cfg: Cfg, sd, name: str, x, ctx: Ctx):
    """EncoderImg.forward (ConvNetworksImgMimic.py:29-36): x [B,1,S,S] -> (mu, logvar) [B,D]."""
    p = f"{name}.feature_extractor"
    # ctx.bf16: the single-channel stem runs on fp32 pixels and fp32 taps (streaming kernel); its output is stored bf16
    h = _q(F.conv2d(x, sd[p + ".conv1.weight"], None, stride=2, padding=1), ctx)
    for i, (_ci, _co, s, pd) in enumerate(img_enc_blocks(cfg)):
        h = _resblock(sd, f"{p}.resblock_{i + 1}.0", h, ctx, stride=s, pad=pd, transposed=False,
                      twod=True, short_name="downsample")
    return _compress(sd, f"{name}.feature_compressor", h, ctx)


def decode_img(cfg: Cfg, sd, name: str, z, ctx: Ctx):
    """DecoderImg.forward (ConvNetworksImgMimic.py:46-54): z [B,D] -> img_hat [B,1,S,S]."""
    h = _q(F.linear(_q(z, ctx), _qw(sd[f"{name}.feature_generator.weight"], ctx), sd[f"{name}.feature_generator.bias"]), ctx)
    h = h.view(h.shape[0], h.shape[1], 1, 1)
    p = f"{name}.img_generator.generator"
    blocks = img_dec_blocks(cfg)
    for i, (_ci, _co, s, pd) in enumerate(blocks):
        h = _resblock(sd, f"{p}.{i}.0", h, ctx, stride=s, pad=pd, transposed=True, twod=True,
                      short_name="upsample")
    k = len(blocks)
    return F.conv_transpose2d(h, sd[f"{p}.{k}.weight"], sd[f"{p}.{k}.bias"], stride=2, padding=1,
                              output_padding=1)


def encode_text(cfg: Cfg, sd, x_ids, ctx: Ctx):
    """EncoderText.forward (ConvNetworksTextMimic.py:23-36): float ids [B,L] -> (mu, logvar)."""
    p = "encoder_text.feature_extractor"
    if cfg.text_encoding == "char":   # char_encoding/FeatureExtractorText.py:70-80: x [B, L, num_features]
        h = x_ids
    else:
        h = _q(F.embedding(x_ids.long(), sd[p + ".embedding.weight"], padding_idx=0), ctx)
    h = h.transpose(-2, -1)
    h = _q(F.conv1d(h, _qw(sd[p + ".conv1.weight"], ctx), sd[p + ".conv1.bias"], stride=2, padding=1), ctx)
    blocks = text_enc_blocks(cfg)
    n_run = 8 if (cfg.len_sequence > 500 or cfg.text_encoding == "char") else 6  # mmvae_text_enc.py:82-84
    for i in range(n_run):
        _ci, _co, s, pd = blocks[i]
        h = _resblock(sd, f"{p}.resblock_{i + 1}.0", h, ctx, stride=s, pad=pd, transposed=False,
                      twod=False, short_name="downsample")
    return _compress(sd, "encoder_text.feature_compressor", h, ctx)


def decode_text(cfg: Cfg, sd, z, ctx: Ctx):
    """DecoderText.forward (ConvNetworksTextMimic.py:51-68): z -> log-probs [B,L,V]."""
    h = _q(F.linear(_q(z, ctx), _qw(sd["decoder_text.feature_generator.weight"], ctx),
                    sd["decoder_text.feature_generator.bias"]), ctx)
    h = h.unsqueeze(-1)
    blocks = text_dec_blocks(cfg)
    if cfg.text_encoding == "char":   # char_encoding/DataGeneratorText.py:51-76
        p = "decoder_text.text_generator"
        for i, (_ci, _co, s, pd) in enumerate(blocks):
            h = _resblock(sd, f"{p}.resblock_{i + 1}.0", h, ctx, stride=s, pad=pd, transposed=True, twod=False,
                          short_name="upsample")
        logits = F.conv_transpose1d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], stride=2, padding=1)
        return F.log_softmax(logits, dim=1).transpose(-2, -1)
    p = "decoder_text.text_generator.generator"
    for i, (_ci, _co, s, pd) in enumerate(blocks):
        h = _resblock(sd, f"{p}.{i}.0", h, ctx, stride=s, pad=pd, transposed=True, twod=False,
                      short_name="upsample")
    k = len(blocks)
    # ctx.bf16: bf16 operands; the logits are a STORED tensor of the bf16 family since round 4 (the head GEMM writes them
    # once, in bf16; log-sum-exp, the target's logit and softmax - onehot are taken from the stored values in fp32), and their
    # gradient re-enters the GEMMs in bf16
    logits = _q(F.conv1d(h, _qw(sd[f"{p}.{k}.weight"], ctx), sd[f"{p}.{k}.bias"]), ctx)
    return F.log_softmax(logits, dim=1).transpose(-2, -1)


# --------------------------------------------------------------------------------------------
# latent space: PoE per subset, KL, mixture selection, reparameterisation
# --------------------------------------------------------------------------------------------
def poe(mus, logvars):
    """mm_div.py:10-17.  mus/logvars: [k,B,D]."""
    var = torch.exp(logvars) + POE_EPS
    T = 1.0 / var
    pd_mu = torch.sum(mus * T, dim=0) / torch.sum(T, dim=0)
    pd_var = 1.0 / torch.sum(T, dim=0)
    return pd_mu, torch.log(pd_var)


def kl_std_normal(mu, logvar, norm):
    """kl_div.py:8-16 with mu1=None."""
    return -0.5 * torch.sum(1 - logvar.exp() - mu.pow(2) + logvar) / float(norm)


def mixture_row_ranges(num_samples: int, k: int) -> List[Tuple[int, int]]:
    """Row range of the batch taken from each of k mixture components (utils.py:55-77 with the
    weights built at BaseMMVae.py:185-187 and re-normalised at :104).  fp32 arithmetic as in torch."""
    w = (1 / float(k)) * torch.ones(k)
    w = w / w.sum()
    ranges, start = [], 0
    for i in range(k):
        if i == k - 1:
            end = num_samples
        else:
            end = start + int(torch.floor(num_samples * w[i]))
        ranges.append((start, end))
        start = end
    return ranges


def fuse_latents(cfg: Cfg, enc: Dict[str, Tuple[torch.Tensor, torch.Tensor]]):
    """BaseMMVae.inference (:139-196) for joint_elbo with whatever modalities ``enc`` holds, plus
    divergence_static_prior (:71-85).  Returns a dict mirroring results['latents'] + divergences."""
    subsets, mus, lvs = {}, [], []
    for key, members in subset_table():
        if all(m in enc for m in members):
            s_mu, s_lv = poe(torch.stack([enc[m][0] for m in members]),
                             torch.stack([enc[m][1] for m in members]))
            subsets[key] = (s_mu, s_lv)
            mus.append(s_mu)
            lvs.append(s_lv)
    mus, lvs = torch.stack(mus), torch.stack(lvs)
    k, nrow = mus.shape[0], mus.shape[1]
    ranges = mixture_row_ranges(nrow, k)
    joint_mu = torch.cat([mus[i, a:b] for i, (a, b) in enumerate(ranges)])
    joint_lv = torch.cat([lvs[i, a:b] for i, (a, b) in enumerate(ranges)])
    weights = ((1 / float(k)) * torch.ones(k)).to(mus.dtype)
    w = weights / weights.sum()
    w = w / w.sum()  # BaseMMVae.py:74-75 reweights the clone again
    klds = torch.stack([kl_std_normal(mus[i], lvs[i], cfg.batch_size) for i in range(k)])
    joint_div = (w * klds).sum(dim=0)
    return {"subsets": subsets, "mus": mus, "logvars": lvs, "weights": weights, "joint": (joint_mu, joint_lv),
            "individual_divs": klds, "joint_divergence": joint_div, "ranges": ranges}


def reparameterize(mu, logvar, eps):
    """utils.py:45-48 with the noise passed in."""
    return eps * torch.exp(0.5 * logvar) + mu


def laplace_nll(x_hat, target, norm):
    """-sum log Laplace(target; loc=x_hat, scale=.75) / norm  (Modality.py:25-30, losses.py:17)."""
    lp = -math.log(2 * LAPLACE_SCALE) - torch.abs(target - x_hat) / LAPLACE_SCALE
    return -lp.sum() / norm


def categorical_nll(logp, target_ids, norm):
    """-sum onehot(target)*normalised(logits) / norm (MimicText.py:37-40; OneHotCategorical
    re-normalises its logits: logits - logsumexp(logits))."""
    logp = logp - torch.logsumexp(logp, dim=-1, keepdim=True)
    if target_ids.dim() == logp.dim():   # char encoding: the target is a [B, L, num_features] one-hot tensor:
        return -(target_ids * logp).sum() / norm   # OneHotCategorical.log_prob = sum(target * log p)
    picked = torch.gather(logp, -1, target_ids.long().unsqueeze(-1))
    return -picked.sum() / norm


# --------------------------------------------------------------------------------------------
# whole step
# --------------------------------------------------------------------------------------------
def forward_step(cfg: Cfg, sd, batch: Dict[str, torch.Tensor], eps: torch.Tensor, ctx: Ctx):
    """basic_routine_epoch (run_epochs.py:52-96) for method=joint_elbo, factorized_representation=False.
    ``batch`` may hold any non-empty subset of modalities for the latent part, but the loss needs all
    three (VAEtrimodalMimic.py:45-46)."""
    enc = {}
    if "PA" in batch:
        enc["PA"] = encode_img(cfg, sd, "encoder_pa", batch["PA"], ctx)
    if "Lateral" in batch:
        enc["Lateral"] = encode_img(cfg, sd, "encoder_lat", batch["Lateral"], ctx)
    if "text" in batch:
        enc["text"] = encode_text(cfg, sd, batch["text"], ctx)
    lat = fuse_latents(cfg, enc)
    z = reparameterize(lat["joint"][0], lat["joint"][1], eps)
    out = {"enc": enc, "latents": lat, "z": z}
    if all(m in batch for m in MOD_ORDER):
        rec = {"PA": decode_img(cfg, sd, "decoder_pa", z, ctx),
               "Lateral": decode_img(cfg, sd, "decoder_lat", z, ctx),
               "text": decode_text(cfg, sd, z, ctx)}
        nll = {"PA": laplace_nll(rec["PA"], batch["PA"], cfg.batch_size),
               "Lateral": laplace_nll(rec["Lateral"], batch["Lateral"], cfg.batch_size),
               "text": categorical_nll(rec["text"], batch["text"], cfg.batch_size)}
        weighted = sum(cfg.rec_weights[m] * nll[m] for m in MOD_ORDER)
        total = weighted + cfg.beta * (cfg.beta_content * lat["joint_divergence"])
        klds = {k: kl_std_normal(mu, lv, cfg.batch_size) for k, (mu, lv) in lat["subsets"].items()}
        out.update(rec=rec, log_probs=nll, total_loss=total, klds=klds)
    return out


# --------------------------------------------------------------------------------------------
# importance-sampled log-likelihood estimator (evaluation)
# --------------------------------------------------------------------------------------------
LOG2PI = float(math.log(2.0 * math.pi))


def log_mean_exp(x, dim=1):
    """mimic/utils/likelihood.py:41-53."""
    m = torch.max(x, dim=dim, keepdim=True)[0]
    return m + torch.log(torch.mean(torch.exp(x - m), dim=dim, keepdim=True))


def gaussian_log_pdf(x, mu, logvar):
    """mimic/utils/likelihood.py:56-67."""
    return torch.sum(-0.5 * LOG2PI - logvar / 2. - torch.pow(x - mu, 2) / (2. * torch.exp(logvar)), dim=1)


def unit_gaussian_log_pdf(x):
    """mimic/utils/likelihood.py:70-80."""
    return torch.sum(-0.5 * LOG2PI - torch.pow(x, 2) / 2., dim=1)


def likelihood_estimates(cfg: Cfg, sd, batch, subset_posterior, eps, scale: float = 0.75):
    """calc_log_likelihood_batch (mimic/evaluation/eval_metrics/likelihood.py:17-96) for
    factorized_representation=False, followed by log_marginal_estimate / log_joint_estimate
    (mimic/utils/likelihood.py:83-147,150-220).

    subset_posterior = (mu, logvar) [B,D] of one subset; eps [K,B,D] the noise get_latent_samples draws
    (likelihood.py:13-18: the posterior is repeated K times sample-major and reparameterised).  Decoders run in eval mode.
    Returns {'PA','Lateral','text','joint'} scalars.  The reference flattens the K x B weights sample-major and then
    views them as (batch_size, n_samples) (likelihood.py:140,217), i.e. row i of the log-mean-exp holds flat elements
    i*K .. i*K+K-1; that grouping is reproduced here."""
    mu, lv = subset_posterior
    k, b = eps.shape[0], mu.shape[0]
    mu_r = mu.unsqueeze(0).repeat(k, 1, 1).view(k * b, -1)
    lv_r = lv.unsqueeze(0).repeat(k, 1, 1).view(k * b, -1)
    z = (eps.view(k * b, -1) * torch.exp(0.5 * lv_r) + mu_r)
    ctx = Ctx("eval")
    rec = {"PA": decode_img(cfg, sd, "decoder_pa", z, ctx), "Lateral": decode_img(cfg, sd, "decoder_lat", z, ctx)}
    logp_text = decode_text(cfg, sd, z, ctx)                                     # [K*B, L, V] log-probabilities
    log_px = {}
    for m in ("PA", "Lateral"):
        tgt = batch[m].unsqueeze(0).repeat(k, 1, 1, 1, 1).view(k * b, *batch[m].shape[1:])
        lp = -math.log(2 * scale) - torch.abs(tgt - rec[m]) / scale           # Laplace(loc, 0.75).log_prob
        log_px[m] = lp.view(k * b, -1).sum(dim=1)
    # OneHotCategorical(logits).log_prob(one_hot) = sum over the vocabulary of one_hot * normalised logits
    norm_logits = logp_text - torch.logsumexp(logp_text, dim=-1, keepdim=True)
    if cfg.text_encoding == "char":   # the [B, L, num_features] target enters as it is (likelihood.py:103-104)
        tgt = batch["text"].unsqueeze(0).repeat(k, 1, 1, 1).view(k * b, *batch["text"].shape[1:])
        log_px["text"] = (tgt * norm_logits).sum(-1).sum(dim=1)
    else:
        ids = batch["text"].long().unsqueeze(0).repeat(k, 1, 1).view(k * b, -1)
        log_px["text"] = norm_logits.gather(-1, ids.unsqueeze(-1)).squeeze(-1).sum(dim=1)
    log_q = gaussian_log_pdf(z, mu_r, lv_r)
    log_pz = unit_gaussian_log_pdf(z)
    out = {}
    for m in ("PA", "Lateral", "text"):
        w = (log_px[m] + log_pz - log_q).view(b, k)
        out[m] = torch.mean(log_mean_exp(w, dim=1))
    w = (log_px["PA"] + log_px["Lateral"] + log_px["text"] + log_pz - log_q).view(b, k)
    out["joint"] = torch.mean(log_mean_exp(w, dim=1))
    return out


def leaf_state(sd, dtype=None):
    """Clone a state_dict into autograd leaves (float tensors) for gradient checks."""
    out = {}
    for k, v in sd.items():
        if v.is_floating_point():
            t = v.detach().clone()
            if dtype is not None:
                t = t.to(dtype)
            t.requires_grad_(not (k.endswith("running_mean") or k.endswith("running_var")))
            out[k] = t
        else:
            out[k] = v.clone()
    return out


def synthetic_batch(cfg: Cfg, nrow: int, seed: int):
    """Mimic_testing-style inputs (dataio/MimicDataset.py:414-428): U[0,1) images, integer ids as fp32.
    Images are quantised to k/255 so fixtures can store them exactly as uint8."""
    g = torch.Generator().manual_seed(seed)
    s = cfg.img_size
    pa = torch.randint(0, 256, (nrow, 1, s, s), generator=g).float() / 255.0
    lat = torch.randint(0, 256, (nrow, 1, s, s), generator=g).float() / 255.0
    if cfg.text_encoding == "char":
        # one-hot characters [L, num_features], as utils/text.py:13-34 (one_hot_encode) produces on real reports.  (The
        # reference's Mimic_testing draws dense uniform noise here, MimicDataset.py:416-417, which current torch's
        # OneHotCategorical.log_prob rejects as outside its support.)
        ids = torch.randint(0, cfg.num_features, (nrow, cfg.len_sequence), generator=g)
        text = F.one_hot(ids, cfg.num_features).float()
    else:
        text = torch.randint(0, cfg.vocab_size, (nrow, cfg.len_sequence), generator=g).float()
    eps = torch.randn((nrow, cfg.class_dim), generator=g)
    return {"PA": pa, "Lateral": lat, "text": text}, eps


def adam_train_step(cfg: Cfg, sd_leaf, opt, batch, eps, ctx: Ctx):
    """One optimiser step of run_epochs.train (:118-131): zero_grad, forward, backward, Adam."""
    opt.zero_grad()
    out = forward_step(cfg, sd_leaf, batch, eps, ctx)
    out["total_loss"].backward()
    opt.step()
    with torch.no_grad():
        for k, v in ctx.new_running.items():
            sd_leaf[k].copy_(v)
    return out
