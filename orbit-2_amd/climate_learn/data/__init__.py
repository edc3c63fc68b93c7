from .itermodule import IterDataModule, SyntheticGridDataModule
