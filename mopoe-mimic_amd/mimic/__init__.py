"""`mimic` -> `mimic_amd` alias package: with `mopoe-mimic_amd/` on sys.path in place of the reference checkout, the
reference's own import lines (`from mimic.networks.VAEtrimodalMimic import VAEtrimodalMimic`,
`from mimic.run_epochs import run_epochs`, `from mimic.utils.experiment import MimicExperiment`, ...) resolve to the
SAME module objects as their `mimic_amd.*` spellings (no second copy of any module state)."""
import importlib
import pkgutil
import sys

import mimic_amd

for _info in pkgutil.walk_packages(mimic_amd.__path__, "mimic_amd."):
    importlib.import_module(_info.name)
for _name, _mod in list(sys.modules.items()):
    if _name == "mimic_amd" or _name.startswith("mimic_amd."):
        sys.modules["mimic" + _name[len("mimic_amd"):]] = _mod
