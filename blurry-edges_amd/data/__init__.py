from .dataset import ShapeDataset, TestDataset
