"""``from model import ...`` of the reference notebooks (pyfiles/model.py) -> srgan_amd.model."""
from srgan_amd.model import *  # noqa: F401,F403
from srgan_amd.model import MinMax  # noqa: F401  (the notebooks import it from ``model``; it arrives there via ``from util import *``)
