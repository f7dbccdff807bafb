"""Modality plugin API and the fused likelihood objects.

Keeps the reference's Modality contract (mimic/modalities/Modality.py:15-47, MimicPA.py:7-17,
MimicLateral.py:7-17, MimicText.py:12-40, modalities/utils.py:4-15): a modality carries ``name``,
``likelihood_name``, ``data_size``, ``gen_quality_eval``, ``file_suffix``, ``encoder``, ``decoder``,
``likelihood`` and ``calc_log_prob(out_dist, target, norm_value)``.  ``likelihood`` constructs an
object with ``.log_prob(target)`` and ``.mean`` like the torch.distributions classes the reference
uses, plus a fused path: summed log-probability in one HIP reduction (no [B,L,V] one-hot, no
elementwise log_prob tensor).
"""
from __future__ import annotations

import math
from abc import ABC, abstractmethod

import torch

from . import ops


# ---- fused likelihoods -------------------------------------------------------------------------
class _LaplaceNll(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loc, target, scale, norm):
        loc_c, tgt = loc.contiguous(), target.contiguous()
        ctx.save_for_backward(loc_c, tgt)
        ctx.scale, ctx.norm = scale, norm
        return ops.laplace_nll_fwd(loc_c, tgt, scale, norm)

    @staticmethod
    def backward(ctx, g):
        loc, tgt = ctx.saved_tensors
        return ops.laplace_nll_bwd(loc, tgt, g.contiguous(), ctx.scale, ctx.norm), None, None, None


class HeadCtx:
    """Side channel between the text decoder and the token likelihood.  The gradient of the token NLL with respect to the
    log-probabilities is one non-zero per row; written out it is a [B, L, V] tensor (115 MB at C2, 460 MB at C3) plus its
    memset, only to be read back by the log-softmax backward.  When the decoder has left a HeadCtx on its output, the
    likelihood's backward hands over (ids, upstream gradient, norm) here and returns a stride-0 ZERO tensor as the dense
    gradient; the decoder's backward then makes the logits' gradient in one pass (ops.token_softmax_grad) and adds
    whatever dense gradient other consumers of the log-probabilities may have produced."""

    def __init__(self):
        self.pending = None    # (ids, g, norm) of the backward in flight


_zero_scalars = {}


def zero_placeholder(shape, device, dtype=torch.float32):
    """a zero tensor of `shape` that occupies one element (stride 0); is_zero_placeholder() recognises it.  `dtype` must be the
    dtype of the tensor whose gradient it stands for: autograd casts a gradient of another dtype -- which would materialise
    the whole [B, L, V] tensor this placeholder exists to avoid"""
    key = (device.type, device.index, dtype)
    z = _zero_scalars.get(key)
    if z is None:
        z = _zero_scalars[key] = torch.zeros(1, dtype=dtype, device=device)
    return z.expand(shape)


def is_zero_placeholder(t) -> bool:
    z = _zero_scalars.get((t.device.type, t.device.index, t.dtype))
    return z is not None and t.data_ptr() == z.data_ptr() and all(s == 0 for s in t.stride())


class _TokenNll(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, ids, norm):
        ids_c = ids.contiguous()
        ctx.save_for_backward(ids_c)
        ctx.shape, ctx.norm = tuple(logp.shape), norm
        ctx.head = getattr(logp, "_mopoe_head_ctx", None)
        return ops.token_nll_fwd(logp.contiguous(), ids_c, norm)

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        if ctx.head is not None and ctx.head.pending is None:
            ctx.head.pending = (ids, g.contiguous(), ctx.norm)
            return zero_placeholder(ctx.shape, g.device), None, None
        return ops.token_nll_bwd(ids, g.contiguous(), ctx.shape, ctx.norm), None, None


class LogitsWithLse:
    """What the word text decoder hands to the likelihood on the model's own path (nets.DecoderText, lazy head): the head's
    LOGITS [B, L, Vpad] in the family's storage type plus their row log-sum-exp [B, L] (fp32) -- the reference's
    LogSoftmax output (word_encoding/DataGeneratorText.py:76-77) in factored form.  The token likelihood and its gradient
    need only these (ops.token_nll_logits_fwd / token_softmax_grad_logits); the dense [B, L, V] fp32 log-probabilities are
    made on demand (`dense()`: evaluation, generation, the likelihood estimator, tests)."""

    def __init__(self, logits_pad, lse, vocab, head_ctx=None):
        self.logits_pad, self.lse, self.vocab, self.head_ctx = logits_pad, lse, vocab, head_ctx
        self._dense = None

    @property
    def shape(self):
        return torch.Size((*self.logits_pad.shape[:-1], self.vocab))

    def dim(self):
        return self.logits_pad.dim()

    def dense_padded(self):
        """fp32 log-probabilities over the padded vocabulary (pad columns ~ -1e30), connected to autograd"""
        if self._dense is None:
            self._dense = self.logits_pad.float() - self.lse.unsqueeze(-1)
        return self._dense

    def dense(self):
        lp = self.dense_padded()
        if lp.shape[-1] == self.vocab:
            return lp
        out = lp[..., :self.vocab]
        out._mopoe_padded = lp
        return out


class _TokenNllLogits(torch.autograd.Function):
    """token NLL from (logits, row log-sum-exp): sum_r (lse_r - logits[r, id_r]) / norm"""

    @staticmethod
    def forward(ctx, logits, lse, ids, norm, head):
        ids_c = ids.contiguous()
        ctx.save_for_backward(ids_c)
        ctx.shape, ctx.norm, ctx.head, ctx.dtype = tuple(logits.shape), norm, head, logits.dtype
        return ops.token_nll_logits_fwd(logits, lse, ids_c, norm)

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        if ctx.head is not None and ctx.head.pending is None:
            # the decoder's backward makes the logits' gradient in one pass from what is handed over here
            ctx.head.pending = (ids, g.contiguous(), ctx.norm)
            return (zero_placeholder(ctx.shape, g.device, ctx.dtype), zero_placeholder(ctx.shape[:-1], g.device), None, None, None)
        # general form (a second likelihood on the same decoder output): d/dlogits = -onehot g / norm, d/dlse = g / norm
        c = g.reshape(()) / ctx.norm
        return (ops.token_nll_bwd(ids, g.contiguous(), ctx.shape, ctx.norm).to(ctx.dtype), c.expand(ctx.shape[:-1]).contiguous(),
                None, None, None)


class _DenseNll(torch.autograd.Function):
    """-sum(target * logp) / norm for a dense / one-hot target of logp's shape (char text encoding)"""

    @staticmethod
    def forward(ctx, logp, target, norm):
        tgt = target.contiguous()
        ctx.save_for_backward(tgt)
        ctx.norm = norm
        return ops.dense_nll_fwd(logp.contiguous(), tgt, norm)

    @staticmethod
    def backward(ctx, g):
        (tgt,) = ctx.saved_tensors
        return ops.dense_nll_bwd(tgt, g.contiguous(), ctx.norm), None, None


class FusedLaplace:
    """Stand-in for torch.distributions.Laplace(loc, scale) as the image decoders' output."""

    def __init__(self, loc, scale):
        self.loc = loc
        const = getattr(scale, "_mopoe_const", None)
        self.scale_value = float(scale) if const is None else const
        self.scale = scale

    @property
    def mean(self):
        return self.loc

    def log_prob(self, value):  # elementwise; evaluation-only path (torch on device)
        return -math.log(2 * self.scale_value) - torch.abs(value - self.loc) / self.scale_value

    def summed_nll(self, target, norm_value):
        """-sum(log_prob(target)) / norm_value as one HIP reduction (0-dim tensor): what the loss adds up"""
        return _LaplaceNll.apply(self.loc, target, self.scale_value, float(norm_value)).view(())

    def summed_log_prob(self, target, norm_value):
        """sum(log_prob(target)) / norm_value (0-dim tensor)."""
        return -self.summed_nll(target, norm_value)

    def log_prob_rows(self, target):
        """log_prob summed per row of loc [R, ...] against target [B, ...] repeated R/B times (row r <-> r % B):
        what the likelihood estimator needs from `log_prob(x_rep).view(R, -1).sum(dim=1)`, without the repeat or
        the elementwise tensor (evaluation only: no gradient)."""
        return ops.laplace_logprob_rows(self.loc.detach().contiguous(), target.contiguous(), self.scale_value)


class FusedOneHotCategorical:
    """Stand-in for torch.distributions.OneHotCategorical(logits=log-probabilities [B,L,V]).  `logits` is either that dense
    tensor or a LogitsWithLse (the word decoder's factored output): `.logits` then materialises the dense tensor on demand."""

    def __init__(self, probs=None, logits=None):
        if logits is None:
            raise NotImplementedError("construct with logits= (the text decoder emits log-probabilities)")
        self._lazy = logits if isinstance(logits, LogitsWithLse) else None
        self._logits = None if self._lazy is not None else logits

    @property
    def logits(self):
        return self._lazy.dense() if self._lazy is not None else self._logits

    @property
    def logits_dim(self):
        """number of dimensions of the log-probability tensor, without materialising it"""
        return self._lazy.dim() if self._lazy is not None else self._logits.dim()

    @property
    def probs(self):
        return torch.exp(self.logits)

    @property
    def mean(self):
        return self.probs

    def log_prob(self, value):  # value: one-hot [B,L,V]; evaluation-only path
        return (value * self.logits).sum(-1)

    def summed_nll(self, target_ids, norm_value):
        """target_ids: float-encoded token ids [B,L] (no one-hot is ever materialised).  The text decoder hands over a
        [..., :V] view of a contiguous tensor padded along V (pad log-probabilities = -1e30): the reductions index the
        padded tensor directly, so neither a compaction copy nor a slice-gradient pass exists."""
        if target_ids.dim() == self.logits_dim:   # char encoding: the [B, L, num_features] one-hot tensor itself
            return _DenseNll.apply(self.logits, target_ids, float(norm_value)).view(())
        if self._lazy is not None:                # word encoding, factored head: no log-softmax tensor at all
            z = self._lazy
            return _TokenNllLogits.apply(z.logits_pad, z.lse, target_ids, float(norm_value), z.head_ctx).view(())
        lp = getattr(self.logits, "_mopoe_padded", self.logits)
        return _TokenNll.apply(lp, target_ids, float(norm_value)).view(())

    def summed_log_prob(self, target_ids, norm_value):
        return -self.summed_nll(target_ids, norm_value)

    def log_prob_rows(self, target_ids):
        """per-row sum over the sequence of the picked log-probabilities: logits [R,L,V] against float ids [B,L]
        repeated R/B times (row r <-> r % B); evaluation only."""
        if target_ids.dim() == self.logits_dim:   # char encoding: dense [B, L, num_features] target
            return ops.dense_logprob_rows(self.logits.detach().contiguous(), target_ids.contiguous())
        lp = self._lazy.dense_padded() if self._lazy is not None else getattr(self.logits, "_mopoe_padded", self.logits)
        return ops.token_logprob_rows(lp.detach().contiguous(), target_ids.contiguous())


def get_likelihood(name: str):
    if name == "laplace":
        return FusedLaplace
    if name == "categorical":
        return FusedOneHotCategorical
    if name in ("bernoulli", "normal"):
        raise NotImplementedError(f"likelihood '{name}' is never selected by the MIMIC modalities "
                                  "(SURVEY §0) and has no HIP path")
    print("likelihood not implemented")
    return None


# ---- modalities --------------------------------------------------------------------------------
class Modality(ABC):
    @abstractmethod
    def save_data(self, exp, d, fn, args):
        ...

    @abstractmethod
    def plot_data(self, exp, d):
        ...

    def calc_log_prob(self, out_dist, target: torch.Tensor, norm_value: int):
        """log P(target | out_dist) summed over everything, divided by norm_value."""
        if hasattr(out_dist, "summed_log_prob"):
            return out_dist.summed_log_prob(target, norm_value)
        return out_dist.log_prob(target).sum() / norm_value

    def calc_nll(self, out_dist, target: torch.Tensor, norm_value: int):
        """-calc_log_prob without the two negations (losses.calc_log_probs negates what calc_log_prob returns)"""
        if hasattr(out_dist, "summed_nll"):
            return out_dist.summed_nll(target, norm_value)
        return -self.calc_log_prob(out_dist, target, norm_value)


class ModalityIMG(Modality):
    def __init__(self, data_size):
        self.data_size = data_size

    def save_data(self, exp, d, fn, args):
        raise NotImplementedError("sample dumps are outside the training hot path (SURVEY §2.1-17)")

    def plot_data(self, exp, d, log_tag=None):
        raise NotImplementedError("plotting is outside the training hot path (SURVEY §2.1-17)")


class _MimicImage(ModalityIMG):
    def __init__(self, name, enc, dec, args):
        self.name = name
        self.likelihood_name = "laplace"
        super().__init__(torch.Size((1, args.img_size, args.img_size)))
        self.gen_quality_eval = True
        self.file_suffix = ".png"
        self.encoder, self.decoder = enc, dec
        self.likelihood = get_likelihood(self.likelihood_name)


class MimicPA(_MimicImage):
    def __init__(self, enc, dec, args):
        super().__init__("PA", enc, dec, args)


class MimicLateral(_MimicImage):
    def __init__(self, enc, dec, args):
        super().__init__("Lateral", enc, dec, args)


class MimicText(Modality):
    def __init__(self, enc, dec, len_sequence, plotImgSize, font, args):
        self.name = "text"
        self.args = args
        self.likelihood_name = "categorical"
        self.len_sequence = len_sequence
        if args.text_encoding == "char":     # MimicText.py:19-21 (only the LENGTH of the alphabet enters the arithmetic)
            self.alphabet = getattr(args, "alphabet", None)
            nf = len(self.alphabet) if self.alphabet else int(args.num_features)
            self.data_size = torch.Size((nf, len_sequence))
        elif args.text_encoding == "word":
            self.data_size = torch.Size((args.vocab_size, len_sequence))
        else:
            raise ValueError(f"text_encoding must be 'word' or 'char', not {args.text_encoding!r}")
        self.plot_img_size, self.font = plotImgSize, font
        self.gen_quality_eval = False
        self.file_suffix = ".txt"
        self.encoder, self.decoder = enc, dec
        self.likelihood = get_likelihood(self.likelihood_name)

    def save_data(self, exp, d, fn, args):
        raise NotImplementedError("sample dumps are outside the training hot path (SURVEY §2.1-17)")

    def plot_data(self, exp, d, log_tag=None):
        raise NotImplementedError("plotting is outside the training hot path (SURVEY §2.1-17)")

    def calc_log_prob(self, out_dist, target: torch.Tensor, norm_value: int):
        # reference MimicText.py:37-40: word targets are one-hot encoded first, char targets ARE [B, L, num_features]
        if self.args.text_encoding == "char":
            if hasattr(out_dist, "summed_log_prob"):
                return out_dist.summed_log_prob(target, norm_value)
            return out_dist.log_prob(target).sum() / norm_value
        if hasattr(out_dist, "summed_log_prob") and target.dim() == out_dist.logits_dim - 1:
            return out_dist.summed_log_prob(target, norm_value)  # float ids, as the data loader yields them
        onehot = torch.nn.functional.one_hot(target.to(torch.int64), num_classes=self.args.vocab_size)
        return out_dist.log_prob(onehot).sum() / norm_value

    def calc_nll(self, out_dist, target: torch.Tensor, norm_value: int):
        fused = hasattr(out_dist, "summed_nll") and (self.args.text_encoding == "char"
                                                     or target.dim() == out_dist.logits_dim - 1)
        return out_dist.summed_nll(target, norm_value) if fused else -self.calc_log_prob(out_dist, target, norm_value)
