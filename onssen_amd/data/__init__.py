from .synthetic_wsj0_2mix import wsj0_2mix_dataloader, SyntheticWsj02mix

__all__ = ["wsj0_2mix_dataloader", "SyntheticWsj02mix"]
