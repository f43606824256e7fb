"""Drop-in for the reference's `dataset` module name (diffusion_training.py:13 `import dataset`): the MRI slice loader."""
from anoddpm_amd.dataset import MRIDataset, cycle, init_dataset_loader  # noqa: F401


def init_datasets(ROOT_DIR, args):
    """dataset.py:351-358"""
    return (MRIDataset(ROOT_DIR=f'{ROOT_DIR}DATASETS/Train/', img_size=args['img_size'], random_slice=args['random_slice']),
            MRIDataset(ROOT_DIR=f'{ROOT_DIR}DATASETS/Test/', img_size=args['img_size'], random_slice=args['random_slice']))
