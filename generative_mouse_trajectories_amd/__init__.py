"""MI355X-native (gfx950) implementation of the WGAN-GP trajectory train step of
jrcalgo/generative-mouse-trajectories, behind the reference's own Python surface
(python/learning/RecurrentGAN.py): Config, Generator, Discriminator, Trainer,
compute_kinematics, trajectory_to_absolute, compute_gradient_penalty,
MouseTrajectoryDataset, collate_trajectories."""
from .config import Config  # noqa: F401
