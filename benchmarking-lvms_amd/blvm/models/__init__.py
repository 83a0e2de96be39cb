from blvm.models.base_model import BaseModel, load_model  # noqa: F401
from blvm.models.lstm import LSTMAudio  # noqa: F401
from blvm.models.srnn import SRNN, SRNNAudio  # noqa: F401
from blvm.models.vrnn import VRNN, VRNNAudio, VRNNCell  # noqa: F401
from blvm.models.wavenet import *  # noqa: F401,F403,E402
from blvm.models.clockwork_vae.clockwork_vae import CWVAE, CWVAEAudio  # noqa: F401,E402
from blvm.models.stcn.stcn import STCN  # noqa: F401,E402
