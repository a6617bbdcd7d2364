from .pretrain_model import PretrainModel

LIGHTNING_MODULES = {
    "pretrain": PretrainModel,
}
