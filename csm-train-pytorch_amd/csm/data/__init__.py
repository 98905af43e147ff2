"""Data loading and processing for CSM (mirror of reference ``src/csm/data/__init__.py``)."""
from .training_data import (
    TrainingExample,
    CSMDataProcessor,
    ContextualExampleGenerator,
    CSMDataset,
    LengthBucketSampler,
    create_dataloader,
    collate_variable_length,
    load_audio,
    resample,
)
from .synthetic import SyntheticCSMDataset

__all__ = ["TrainingExample", "CSMDataProcessor", "ContextualExampleGenerator", "CSMDataset", "LengthBucketSampler",
           "create_dataloader", "collate_variable_length", "load_audio", "resample", "SyntheticCSMDataset"]
