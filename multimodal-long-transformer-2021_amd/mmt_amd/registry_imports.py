"""`src/registry_imports.py`: importing this module registers the experiments
(`mmt/pretraining`, `mmt/classification`, `mmt/retrieval`) and the task classes."""
from . import configs, tasks  # noqa: F401
