"""``from util import ...`` of the reference notebooks (pyfiles/util.py) -> srgan_amd."""
from srgan_amd.inference import cuda2cpu, cuda2numpy, image_from_output  # noqa: F401
from srgan_amd.losses import (GaussianHistogram, class_encode, corrcoef, corrcoef_loss, get_domainloss_D,  # noqa: F401
                              get_loss_D, get_target, histogram_imitation, load_classifier, weights_init)
from srgan_amd.data import pickle_load  # noqa: F401
from srgan_amd.model import MinMax  # noqa: F401
