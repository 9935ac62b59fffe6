"""Hyper-parameter bag with the field names and defaults of the reference Config
(RecurrentGAN.py:44-91).  Plain class attributes, instance-mutable, like the reference.
`use_amp` defaults to False here: fp32 is the parity configuration."""
from typing import Tuple


class Config:
    # data
    screen_width: float = 1920.0
    screen_height: float = 1080.0
    min_trajectory_length: int = 5
    max_trajectory_length: int = 200
    # model
    latent_dim: int = 64
    generator_hidden_dim: int = 256
    generator_num_layers: int = 2
    discriminator_hidden_dim: int = 128
    discriminator_num_layers: int = 2
    # training
    batch_size: int = 32
    learning_rate: float = 1e-4
    betas: Tuple[float, float] = (0.5, 0.9)
    n_critic: int = 1
    gradient_penalty_weight: float = 10.0
    gp_every_n: int = 4
    gp_batch_frac: float = 0.25
    use_amp: bool = False
    epochs: int = 1000
    num_workers: int = 4
    pin_memory: bool = True
    patience: int = 50
    lr_patience: int = 20
    lr_factor: float = 0.5
    min_lr: float = 1e-6
    # generation
    distance_threshold: float = 0.02
    max_generation_steps: int = 200
    # teacher forcing
    teacher_forcing_start: float = 1.0
    teacher_forcing_end: float = 0.0
    teacher_forcing_decay_epochs: int = 100
    # loss weights
    endpoint_loss_weight: float = 50.0
    direction_loss_weight: float = 10.0
    # dropout probabilities (hard-coded in the reference: RG.py:456, 726, 733); exposed so
    # that parity runs can zero them exactly as the golden-vector script does
    lstm_dropout: float = 0.1
    head_dropout: float = 0.3
    # arithmetic of the critic's recurrent body: 'fp32' (parity configuration, H <= 256: persistent fp32-MFMA kernels) or
    # 'bf16' (BASELINE.json configs[2]: bf16 operands, fp32 accumulation / state / master weights; H a multiple of 256;
    # ops_bf16).  Not a reference field: the reference reaches reduced precision through torch.autocast (use_amp).
    compute_dtype: str = 'fp32'
