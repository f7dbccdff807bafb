"""Hot-path subset of the reference's MimicExperiment (mimic/utils/experiment.py:41-92,171-190,
BaseExperiment.py:66-82): modalities, subsets, model, optimizer, reconstruction weights.  Datasets,
classifiers, CSV bookkeeping, fonts and plotting are outside the training step."""
from __future__ import annotations

import argparse
import os
from itertools import chain, combinations

import torch
import torch.optim as optim

from ..nets import DecoderImg, DecoderText, EncoderImg, EncoderText
from ..mmvae import VAEtrimodalMimic
from ..plugins import MimicLateral, MimicPA, MimicText


def default_flags(**overrides) -> argparse.Namespace:
    """The flag fields the hot path reads, with the reference's defaults / cluster-config values
    (mimic/utils/flags.py:23-114, BaseFlags.py:4-113, configs/leomed_mimic_config.json)."""
    f = argparse.Namespace(
        img_size=128, image_channels=1, DIM_img=64, DIM_text=128, class_dim=128, batch_size=64,
        style_pa_dim=0, style_lat_dim=0, style_text_dim=0, text_encoding="word", len_sequence=128,
        num_features=71,   # text_encoding='char': alphabet size (experiment.py:49-51); len_sequence is then 1024 (flags.py:157)
        vocab_size=3517, text_gen_lastlayer="softmax", feature_extractor_img="resnet",
        factorized_representation=False, method="joint_elbo", modality_poe=False, modality_moe=False,
        modality_jsd=False, joint_elbo=True, poe_unimodal_elbos=False, only_text_modality=False,
        beta=1.0, beta_style=1.0, beta_content=1.0, beta_m1_style=1.0, beta_m2_style=1.0, beta_m3_style=1.0,
        div_weight_uniform_content=0.25, div_weight_m1_content=0.25, div_weight_m2_content=0.25,
        div_weight_m3_content=0.25, rec_weight_m1=0.33, rec_weight_m2=0.33, rec_weight_m3=0.33,
        initial_learning_rate=5e-4, beta_1=0.9, beta_2=0.999, dataset="testing", distributed=False,
        steps_per_training_epoch=0, seed=0,
        # (evaluated only when the caller passes no device: a launcher process must not initialise the GPU)
        device=overrides["device"] if "device" in overrides else torch.device("cuda" if torch.cuda.is_available() else "cpu"),
        start_epoch=0, end_epoch=1, eval_freq=10, world_size=1, dataloader_workers=0, weighted_sampler=False,
        dir_data=".", word_min_occ=3, undersample_dataset=False, binary_labels=False,   # real tensor datasets (dataio.Mimic)
        compute_dtype="fp32",   # 'bf16': bf16 storage + bf16 MFMA with fp32 accumulation (BASELINE configs #3, #5)
        mm_vae_save="mm_vae", start_early_stopping_epoch=0, max_early_stopping_index=5, testing_batches=2,
        encoder_save_m1="encoderM1", encoder_save_m2="encoderM2", encoder_save_m3="encoderM3",
        decoder_save_m1="decoderM1", decoder_save_m2="decoderM2", decoder_save_m3="decoderM3",
        dir_checkpoints=".")
    f.__dict__.update(overrides)
    f.alpha_modalities = [f.div_weight_uniform_content, f.div_weight_m1_content, f.div_weight_m2_content,
                          f.div_weight_m3_content]
    return f


from ..dataio.utils import get_str_labels  # noqa: E402


class HotPathExperiment:
    """Carries exactly what run_epochs.basic_routine_epoch / train read from the experiment object."""

    def __init__(self, flags):
        self.flags = flags
        self.dataset = flags.dataset
        self.labels = get_str_labels(getattr(flags, "binary_labels", False))
        # (the datasets come first, as in the reference: a real split sets flags.vocab_size / num_features for the networks)
        self.dataset_train, self.dataset_test = self.set_dataset()
        self.modalities = self.set_modalities()
        self.num_modalities = len(self.modalities)
        self.subsets = self.set_subsets()
        self.mm_vae = self.set_model()
        self.optimizer = None
        self.rec_weights = self.set_rec_weights()
        self.style_weights = {"PA": flags.beta_m1_style, "Lateral": flags.beta_m2_style, "text": flags.beta_m3_style}

    def set_dataset(self):
        """mimic/utils/experiment.py:94-111: dataset 'testing' = the synthetic Mimic_testing pair (vocab_size 3517); anything
        else = the tensor files under flags.dir_data (splits 'train' and 'eval'), which also set flags.vocab_size."""
        from ..dataio.MimicDataset import Mimic, Mimic_testing
        if self.flags.dataset == "testing":
            self.flags.vocab_size = getattr(self.flags, "vocab_size", 3517)
            return Mimic_testing(self.flags), Mimic_testing(self.flags)
        if getattr(self.flags, "only_text_modality", False):
            raise NotImplementedError("the text-only model (VAETextMimic) is out of scope (SURVEY 2.1-8)")
        return Mimic(self.flags, self.labels, split="train"), Mimic(self.flags, self.labels, split="eval")

    def set_modalities(self):
        f = self.flags
        mod1 = MimicPA(EncoderImg(f, f.style_pa_dim), DecoderImg(f, f.style_pa_dim), f)
        mod2 = MimicLateral(EncoderImg(f, f.style_lat_dim), DecoderImg(f, f.style_lat_dim), f)
        mod3 = MimicText(EncoderText(f, f.style_text_dim), DecoderText(f, f.style_text_dim), f.len_sequence,
                         None, None, f)
        return {mod1.name: mod1, mod2.name: mod2, mod3.name: mod3}

    def set_subsets(self):
        xs = list(self.modalities)
        subsets = {}
        for names in chain.from_iterable(combinations(xs, n) for n in range(len(xs) + 1)):
            subsets["_".join(sorted(names))] = [self.modalities[m] for m in sorted(names)]
        return subsets

    def set_model(self):
        return VAEtrimodalMimic(self.flags, self.modalities, self.subsets)

    def set_optimizer(self, capturable=None):
        params = list(self.mm_vae.parameters())
        # same Adam arithmetic as the reference's optim.Adam (experiment.py:171-178).  On the GPU the step is
        # mimic_amd.optim.HipAdam: every tensor updated by one kernel family (csrc/adam.hip), step counter AND learning
        # rate on the device, so the step can sit inside a hipGraph (run_epochs.GraphedTrainStep) and a scheduler's lr
        # change (Callbacks' ReduceLROnPlateau fills the tensor in place) reaches the captured optimiser without a
        # re-capture.  MOPOE_TORCH_ADAM=1 (or capturable=False) keeps PyTorch's own fused multi-tensor Adam.
        fused = bool(params) and all(p.is_cuda for p in params)
        if capturable is None:
            capturable = fused
        capturable = bool(capturable and fused)
        lr = self.flags.initial_learning_rate
        betas = (self.flags.beta_1, self.flags.beta_2)
        shadows = [m._shadow for m in self.mm_vae.modules() if getattr(m, "_shadow", None) is not None]
        if capturable and os.environ.get("MOPOE_TORCH_ADAM", "0") != "1":
            self.set_hip_adam(params, lr, betas, shadows)
            return
        for sh in shadows:
            sh.bind_optimizer(False)
        if capturable:
            lr = torch.tensor(float(lr), dtype=torch.float32, device=params[0].device)
        self.optimizer = optim.Adam(params, lr=lr, betas=betas, fused=fused, capturable=capturable)

    def set_hip_adam(self, params, lr, betas, shadows):
        from ..optim import HipAdam
        self.optimizer = HipAdam([p for p in params if p.requires_grad], lr=lr, betas=betas)
        # bf16 family: the kernel that updates a master weight also rewrites its bf16 copy (no cast pass per step)
        lowp = {}
        for sh in shadows:
            lowp.update(sh.bind_optimizer(True))
        if lowp:
            self.optimizer.lowp = [lowp.get(id(p)) for p in self.optimizer._params]

    def set_rec_weights(self):
        f = self.flags
        return {"PA": f.rec_weight_m1, "Lateral": f.rec_weight_m2, "text": f.rec_weight_m3}


MimicExperiment = HotPathExperiment   # the reference's class name (mimic/utils/experiment.py:41)
