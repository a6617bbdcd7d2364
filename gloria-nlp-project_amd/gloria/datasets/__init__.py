from .synthetic import SyntheticPretrainDataModule

# the reference's registry keys (gloria/datasets/__init__.py:5-11); every key maps to the synthetic
# generator here because the real datasets are credentialed and need network access
DATA_MODULES = {
    "imagenome": SyntheticPretrainDataModule,
    "chexpert": SyntheticPretrainDataModule,
    "mimic-cxr": SyntheticPretrainDataModule,
    "synthetic": SyntheticPretrainDataModule,
}
