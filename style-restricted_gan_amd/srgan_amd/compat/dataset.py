"""``from dataset import ...`` of the reference notebooks (pyfiles/dataset.py) -> srgan_amd.data."""
from srgan_amd.data import FaceDataset, get_class_label  # noqa: F401
