"""Same module / class names as the reference's `models/` package, so the reference's scripts
(`from models.clip_wrapper import CLIPWrapper`, reference train.py:3-4) only change their import
root -- or put `tap-clip_amd/` on `sys.path`."""
from .attribution_monitor import AttributionMonitor
from .clip_wrapper import CLIPWrapper
from .model_wrapper import FullModel
from .prompt_adjustor import PromptAdjustor
from .prompt_learner import PromptLearner

__all__ = ["AttributionMonitor", "CLIPWrapper", "FullModel", "PromptAdjustor", "PromptLearner"]
