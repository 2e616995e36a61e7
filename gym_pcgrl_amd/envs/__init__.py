from .batched_env import BatchedPcgrlEnv, InfoBatch
from .pcgrl_env import PcgrlEnv
from .problems import PROBLEMS
from .representations import REPRESENTATIONS

__all__ = ["BatchedPcgrlEnv", "InfoBatch", "PcgrlEnv", "PROBLEMS", "REPRESENTATIONS"]
