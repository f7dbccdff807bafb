"""mimic_amd -- MI355X-native drop-in for the MoPoE-MIMIC joint-ELBO training path.

Mirrors the reference package's module paths for the hot path (``mimic_amd.networks.*``,
``mimic_amd.modalities.*``, ``mimic_amd.utils.*``, ``mimic_amd.evaluation.losses``,
``mimic_amd.run_epochs``); all arithmetic runs in libmopoe_hip.so (csrc/, include/mopoe_hip.h).
"""
__version__ = "0.1.0"
