"""``from util_notebook import ...`` of the reference notebooks (pyfiles/util_notebook.py) -> srgan_amd."""
from srgan_amd.inference import dic_init, get_output_and_plot, get_samples  # noqa: F401
from srgan_amd.trainer import SingleGAN_training, SRGAN_training  # noqa: F401
