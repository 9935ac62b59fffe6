"""MI355X-native (gfx950) implementation of the WGAN-GP trajectory train step of
jrcalgo/generative-mouse-trajectories, behind the reference's own Python surface
(python/learning/RecurrentGAN.py): Config, Generator, Discriminator, Trainer,
compute_kinematics, trajectory_to_absolute, compute_gradient_penalty,
MouseTrajectoryDataset, collate_trajectories.  GPU only; see DESIGN.md."""
from .config import Config  # noqa: F401
from .data import LengthBucketedBatchSampler, MouseTrajectoryDataset, collate_trajectories  # noqa: F401
from .models import Discriminator, Generator, compute_kinematics, trajectory_to_absolute  # noqa: F401
from .trainer import Trainer, compute_gradient_penalty  # noqa: F401
