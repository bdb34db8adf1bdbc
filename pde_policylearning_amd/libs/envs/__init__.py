from .control_env import ChannelFlowRHS  # noqa: F401
