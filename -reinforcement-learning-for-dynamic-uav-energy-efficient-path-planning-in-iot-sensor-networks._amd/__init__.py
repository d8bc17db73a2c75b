"""MI355X-native batched UAV-IoT data-collection environment (one hot path of the reference:
UAVEnvironment.reset()/step(), /root/reference/src/environment/uav_env.py:400-488).

    from uavenv_amd import BatchedUAVEnv, UAVEnvironment, UAVVecEnv

The compute lives in csrc/ (hand-written HIP for gfx950) behind the C ABI of include/uavenv.h;
this package is the Python host side that mirrors the reference's gymnasium / SB3 interfaces.
"""
from . import _native
from ._native import (FLAG_AUTO_RESET, FLAG_FAR_START, FLAG_JAIN_BONUS, FLAG_PROX_SHAPING, FLAG_RANDOM_LAYOUT,
                      UavEnvConfig, UavEnvError, default_config)
from .attention import FusedAttentionFeatures, pack_attention_weights
from .batched_env import BatchedUAVEnv, config_from_kwargs
from .frame_stack import FrameStack
from .gym_env import CURRICULUM_STAGES, DomainRandEnv, UAVEnvironment
from .replay import TransitionRing
from .learner import DQNLearner, QNetwork, REFERENCE_HYPERPARAMS, td_loss
from .vec_env import UAVVecEnv

__all__ = ["DQNLearner", "QNetwork", "REFERENCE_HYPERPARAMS", "td_loss", "BatchedUAVEnv", "UAVEnvironment", "DomainRandEnv", "UAVVecEnv", "TransitionRing", "FrameStack", "FusedAttentionFeatures", "pack_attention_weights", "CURRICULUM_STAGES", "UavEnvConfig", "UavEnvError", "default_config", "config_from_kwargs",
           "FLAG_AUTO_RESET", "FLAG_FAR_START", "FLAG_JAIN_BONUS", "FLAG_PROX_SHAPING", "FLAG_RANDOM_LAYOUT"]
