from .dataset import MultiDomainDataset  # noqa: F401
