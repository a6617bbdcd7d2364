"""gloria - MI355X-native GLoRIA pretraining hot path (drop-in for the reference package's
`gloria.builder` / `gloria.lightning.PretrainModel` / `gloria.loss.gloria_loss` surface)."""

from . import hipgraph  # noqa: F401  (first: sets the HIP runtime's graph flag before the GPU is touched)
from . import loss  # noqa: F401
from . import builder  # noqa: F401
