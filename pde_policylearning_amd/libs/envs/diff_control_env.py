"""libs/envs/diff_control_env.py surface: the PINO loss the observer fine-tuning calls."""
from ..pino_utils.losses import Channelflow_PINO_loss, PINO_loss3d, get_forcing  # noqa: F401
