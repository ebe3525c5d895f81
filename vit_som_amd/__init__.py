"""vit_som_amd: MI355X-native (gfx950) ViT-SOM training step behind the reference's
``ViTSOM`` / ``SOMLayer`` module surface (models/vit_som.py, models/som_layer.py)."""
from . import _lib  # noqa: F401  (fails loudly when libvitsom_hip.so is absent)
from . import ops  # noqa: F401
from .model import FusedAdamW, SOMLayer, ViTAutoencoder, ViTSOM, param_groups_lrd  # noqa: F401,E402
from .desom import DESOM, Autoencoder  # noqa: F401,E402
from . import evaluation  # noqa: F401,E402
