"""Host-side mirror of the reference's model package (same class names, constructor kwargs, forward signatures and
state_dict keys as mattm458/tacotron2 `model/`), backed by the gfx950 HIP library."""
from .tacotron2 import Tacotron2  # noqa: F401
from .tts_model import TTSModel  # noqa: F401
